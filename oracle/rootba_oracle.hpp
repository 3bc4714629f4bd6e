// ============================================================================
// oracle/rootba_oracle.hpp — CPU restatement of the reference's square-root BA
// inner solver (NikolausDemmel/rootba, QR path).
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
// build, link, load or execute anything under oracle/. The product path
// (rootba_amd/, include/) never calls into it and has no CPU fallback.
//
// PARITY STATUS: pinned against the reference's own code, NOT against the
// arithmetic inside its third-party libraries.
// The reference holds no golden vectors for this path and cannot be built with
// its own dependencies here (Eigen 3.4, Sophus, basalt-headers, TBB headers,
// glog, fmt are un-vendored submodules; no network). oracle/build_ref.sh
// compiles the reference's hot-path sources unmodified against stand-ins for
// those libraries (oracle/ref_shims/, oracle/ref_driver.cpp); this restatement
// agrees with that build to 1e-16..1e-14 (double) / ~1e-6 (float) on every
// stage, per solver variant and over whole LM runs
// (tests/test_oracle_vs_reference.py, tests/test_reference_golden.py).
// Eigen's / Sophus' / basalt's own arithmetic is restated in BOTH places from
// the published algorithms - that part remains "parity unpinned" and is bounded
// by the reference's own boundary tests (projection == in-tree Snavely formula,
// analytic == numeric Jacobians), by an independent autograd / dense-normal-
// equation derivation (tests/test_oracle_independent.py) and by the property
// tests the reference uses for itself (tests/test_oracle_*.py).
//
// Every function cites the reference file:line (relative to /root/reference)
// whose behaviour it restates. Third-party arithmetic (Eigen Householder /
// Givens / LLT, Sophus SO3::exp, basalt BalCamera::project) is restated from
// the libraries' published algorithms.
//
// Plain C++17, no dependencies, templated on Scalar in {float,double}.
// Threading: OpenMP parallel-for over landmarks with per-thread accumulators
// (the reference's `reduction_alg = 0` strategy, linearization_qr.hpp:294-334).
// ============================================================================
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#include <memory>
#include <stdexcept>
#include <cstdlib>
#endif

namespace orc {

// ---------------------------------------------------------------------------
// Options: the SolverOptions fields the hot path reads
// (src/rootba/bal/solver_options.hpp:111-281; consumed at
//  src/rootba/solver/linearizor_qr.cpp:58-68, linearizor_base.cpp:87-93,
//  bal_bundle_adjustment.cpp:264-272).
// Layout is mirrored by `orc_options` in oracle_capi.cpp and by `rba_options`
// in include/rootba_hip.h.
// ---------------------------------------------------------------------------
struct Options {
  int use_householder = 1;             // use_householder_marginalization
  int use_valid_projections_only = 0;  // = optimized_cost != ERROR
  int robust_norm = 0;                 // 0 NONE, 1 HUBER
  double huber_parameter = 1.0;
  double jacobi_scaling_eps = 0.0;  // 0 -> Sophus epsilonSqrt<Scalar>
  int preconditioner_type = 1;      // 0 JACOBI, 1 SCHUR_JACOBI, 2 POWER_SCHUR_COMPLEMENT
  int reduction_alg = 1;
  int power_order = 10;
  int min_cg_it = 0;
  int max_cg_it = 500;
  double eta = 0.1;
  int num_threads = 0;  // 0 = all
  // LM driver (bal_bundle_adjustment.cpp:264-272; solver_options.hpp:141-168)
  int max_num_iterations = 20;
  double min_relative_decrease = 0.0;
  double initial_trust_region_radius = 1e4;
  double min_trust_region_radius = 1e-32;
  double max_trust_region_radius = 1e16;
  double function_tolerance = 1e-6;
  double initial_vee = 2.0;
  double vee_factor = 2.0;
  int optimized_cost = 0;  // 0 ERROR, 1 ERROR_VALID, 2 ERROR_VALID_AVG
  int staged_execution = 1;
  // SolverOptions::SolverType (src/rootba/bal/solver_options.hpp:58-62): 0 SQUARE_ROOT (LinearizorQR),
  // 1 SCHUR_COMPLEMENT (LinearizorSC, src/rootba/solver/linearizor_sc.cpp:70-211; the
  // reduced matrix is kept DENSE here, i.e. small problems only)
  int solver_type = 0;
};

// src/rootba/bal/residual_info.hpp:57-96
struct ResidualItem {
  int num_obs = 0;
  double error = 0;
  double residual_sum = 0;
  double error_avg() const { return num_obs > 0 ? error / num_obs : 0.0; }
};
struct ResidualInfo {
  ResidualItem all, valid;
  int is_numerically_valid = 1;
};

// CG summary (src/rootba/cg/conjugate_gradient.hpp:97-108)
struct CgSummary {
  int termination_type = 0;  // 0 NO_CONVERGENCE, 1 SUCCESS, 2 FAILURE
  int num_iterations = 0;
};

// One row of the LM log (subset of IterationSummary,
// src/rootba/solver/solver_summary.hpp:99-204)
struct LmIteration {
  int iteration = 0;
  int step_is_valid = 0;
  int step_is_successful = 0;
  int cg_iterations = 0;
  int cg_termination = 0;
  double cost = 0;        // cost.all.error AFTER this iteration's step attempt
  double cost_valid = 0;  // cost.valid.error
  double lambda = 0;      // damping used for this iteration's solve
  double relative_decrease = 0;
  double l_diff = 0;
  double inc_norm = 0;  // |scaled pose increment| returned by solve()
  double iteration_time = 0;
  double stage1_time = 0, stage2_time = 0, precond_time = 0, pcg_time = 0,
         backsub_time = 0, residual_time = 0;
  // ResidualInfo of the evaluated state (solver_summary.hpp cost.{all,valid})
  int num_obs = 0, num_obs_valid = 0;
  double residual_sum = 0, residual_sum_valid = 0;
};

// Sophus::Constants<Scalar>::epsilonSqrt() (upstream Sophus: epsilon = 1e-10
// double / 1e-5 float). Used at linearizor_base.cpp:72-79 and as projection
// validity threshold (cf. src/rootba/ceres/bal_residuals.hpp:63). Unpinned by
// any reference test: recorded as an assumption.
template <class S>
inline S epsilon() {
  return sizeof(S) == 4 ? S(1e-5) : S(1e-10);
}
template <class S>
inline S epsilon_sqrt() {
  return std::sqrt(epsilon<S>());
}

inline double now_seconds() {
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return 0.0;
#endif
}

// ---------------------------------------------------------------------------
// Geometry (row B, C of SURVEY.md §8a)
// ---------------------------------------------------------------------------

// Sophus SO3::matrix() == Eigen::Quaternion::toRotationMatrix() for the unit
// quaternion stored (x,y,z,w) (bal_problem.hpp:84-95).
template <class S>
inline void quat_to_rot(const S* q, S R[9]) {
  const S x = q[0], y = q[1], z = q[2], w = q[3];
  const S tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const S twx = tx * w, twy = ty * w, twz = tz * w;
  const S txx = tx * x, txy = ty * x, txz = tz * x;
  const S tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1 - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1 - (txx + tyy);
}

// BalBundleAdjustmentHelper::linearize_point
// (src/rootba/bal/bal_bundle_adjustment_helper.cpp:111-149) with
// basalt::BalCamera::project restated per SURVEY.md A.1 (value pinned in-tree
// by src/rootba/bal/snavely_projection.hpp:182-190).
// cam = (qx,qy,qz,qw,tx,ty,tz,f,k1,k2). Jp is 2x6 row-major, Ji 2x3, Jl 2x3.
// Returns projection_valid; when (!ignore_validity_check && !valid) the
// Jacobians are left untouched and false is returned (helper.cpp:131-133).
template <class S>
inline bool linearize_point(const S* obs, const S* p_w, const S* cam,
                            bool ignore_validity_check, S res[2], S* Jp,
                            S* Ji, S* Jl) {
  S R[9];
  quat_to_rot(cam, R);
  const S* t = cam + 4;
  const S f = cam[7], k1 = cam[8], k2 = cam[9];
  const S px = R[0] * p_w[0] + R[1] * p_w[1] + R[2] * p_w[2] + t[0];
  const S py = R[3] * p_w[0] + R[4] * p_w[1] + R[5] * p_w[2] + t[1];
  const S pz = R[6] * p_w[0] + R[7] * p_w[1] + R[8] * p_w[2] + t[2];

  const S mx = px / pz, my = py / pz;
  const S r2 = mx * mx + my * my;
  const S r4 = r2 * r2;
  const S rp = S(1) + k1 * r2 + k2 * r4;
  res[0] = f * mx * rp - obs[0];
  res[1] = f * my * rp - obs[1];
  const bool valid = pz >= epsilon_sqrt<S>();

  if (!ignore_validity_check && !valid) return false;

  if (Jp || Ji || Jl) {
    // d proj / d p_c (2x3; the 4th homogeneous column is zero)
    const S tmp = k1 + S(2) * k2 * r2;
    S J[6];
    J[0] = f * (rp + S(2) * mx * mx * tmp) / pz;
    J[1] = S(2) * f * mx * my * tmp / pz;
    J[2] = -f * mx * (rp + S(2) * r2 * tmp) / pz;
    J[3] = J[1];
    J[4] = f * (rp + S(2) * my * my * tmp) / pz;
    J[5] = -f * my * (rp + S(2) * r2 * tmp) / pz;
    if (Ji) {
      Ji[0] = mx * rp;
      Ji[1] = f * mx * r2;
      Ji[2] = f * mx * r4;
      Ji[3] = my * rp;
      Ji[4] = f * my * r2;
      Ji[5] = f * my * r4;
    }
    if (Jp) {
      // d_res_d_xi = d_res_d_p * [I3 | -hat(p_c)]   (helper.cpp:135-142)
      for (int r = 0; r < 2; ++r) {
        const S j0 = J[3 * r + 0], j1 = J[3 * r + 1], j2 = J[3 * r + 2];
        Jp[6 * r + 0] = j0;
        Jp[6 * r + 1] = j1;
        Jp[6 * r + 2] = j2;
        Jp[6 * r + 3] = -j1 * pz + j2 * py;
        Jp[6 * r + 4] = j0 * pz - j2 * px;
        Jp[6 * r + 5] = -j0 * py + j1 * px;
      }
    }
    if (Jl) {
      // d_res_d_l = d_res_d_p * R   (helper.cpp:144-146)
      for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c)
          Jl[3 * r + c] = J[3 * r + 0] * R[0 + c] + J[3 * r + 1] * R[3 + c] +
                          J[3 * r + 2] * R[6 + c];
    }
  }
  return valid;
}

// compute_error_weight (helper.cpp:43-66)
template <class S>
inline void compute_error_weight(const Options& o, S res_squared, S& error,
                                 S& weight) {
  if (o.robust_norm == 1) {
    const S thresh = S(o.huber_parameter);
    const S w = res_squared < thresh * thresh
                    ? S(1.0)
                    : thresh / std::sqrt(res_squared);
    error = S(0.5) * (S(2) - w) * w * res_squared;
    weight = w;
  } else {
    error = S(0.5) * res_squared;
    weight = S(1.0);
  }
}

// Sophus::SO3::exp(omega) as unit quaternion (x,y,z,w).
template <class S>
inline void so3_exp_quat(const S* w, S q[4]) {
  const S theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  S imag, real;
  if (theta_sq < epsilon<S>() * epsilon<S>()) {
    const S theta_po4 = theta_sq * theta_sq;
    imag = S(0.5) - S(1.0 / 48.0) * theta_sq + S(1.0 / 3840.0) * theta_po4;
    real = S(1) - S(1.0 / 8.0) * theta_sq + S(1.0 / 384.0) * theta_po4;
  } else {
    const S theta = std::sqrt(theta_sq);
    const S half = S(0.5) * theta;
    imag = std::sin(half) / theta;
    real = std::cos(half);
  }
  q[0] = imag * w[0];
  q[1] = imag * w[1];
  q[2] = imag * w[2];
  q[3] = real;
}

// Hamilton product a*b, (x,y,z,w) storage.
template <class S>
inline void quat_mul(const S* a, const S* b, S* o) {
  const S ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const S bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by - ax * bz + ay * bw + az * bx;
  o[2] = aw * bz + ax * by - ay * bx + az * bw;
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
}

// Camera::apply_inc_pose / apply_inc_intrinsics (bal_problem.hpp:97-109):
// T <- se3_expd(inc) * T with se3_expd(v,w) = SE3(SO3::exp(w), v) (decoupled),
// i.e. R <- exp(w) R, t <- exp(w) t + v; intrinsics += inc[6..8].
// Sophus SO3 multiplication renormalises the quaternion to first order
// (SO3::operator*= in upstream Sophus); restated here.
template <class S>
inline void apply_inc_camera(S* cam, const S* inc9) {
  S dq[4];
  so3_exp_quat(inc9 + 3, dq);
  S dR[9];
  quat_to_rot(dq, dR);
  S q[4];
  quat_mul(dq, cam, q);
  // first-order renormalisation as in Sophus::SO3Base::operator*
  const S sn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (sn != S(1)) {
    const S sc = S(2.0) / (S(1.0) + sn);
    for (int i = 0; i < 4; ++i) q[i] *= sc;
  }
  const S tx = cam[4], ty = cam[5], tz = cam[6];
  cam[4] = dR[0] * tx + dR[1] * ty + dR[2] * tz + inc9[0];
  cam[5] = dR[3] * tx + dR[4] * ty + dR[5] * tz + inc9[1];
  cam[6] = dR[6] * tx + dR[7] * ty + dR[8] * tz + inc9[2];
  for (int i = 0; i < 4; ++i) cam[i] = q[i];
  cam[7] += inc9[6];
  cam[8] += inc9[7];
  cam[9] += inc9[8];
}

// ---------------------------------------------------------------------------
// Small dense helpers (Eigen restatements)
// ---------------------------------------------------------------------------

// Eigen::JacobiRotation::makeGivens(p, q) for real scalars (A.4).
template <class S>
inline void make_givens(S p, S q, S& c, S& s) {
  if (q == S(0)) {
    c = p < S(0) ? S(-1) : S(1);
    s = S(0);
  } else if (p == S(0)) {
    c = S(0);
    s = q < S(0) ? S(1) : S(-1);
  } else if (std::abs(p) > std::abs(q)) {
    const S t = q / p;
    S u = std::sqrt(S(1) + t * t);
    if (p < S(0)) u = -u;
    c = S(1) / u;
    s = -t * c;
  } else {
    const S t = p / q;
    S u = std::sqrt(S(1) + t * t);
    if (q < S(0)) u = -u;
    s = -S(1) / u;
    c = -t * s;
  }
}

// MatrixBase::applyOnTheLeft(p, q, G): row_p' = c row_p + s row_q,
// row_q' = -s row_p + c row_q  (A.4).
template <class S>
inline void apply_rot_rows(S* row_p, S* row_q, int n, S c, S s) {
  for (int i = 0; i < n; ++i) {
    const S xi = row_p[i], yi = row_q[i];
    row_p[i] = c * xi + s * yi;
    row_q[i] = -s * xi + c * yi;
  }
}

// 9x9 (n x n) inverse through LLT of the matrix defined by the UPPER triangle
// of `a` (row-major), solve against identity
// (src/rootba/cg/preconditioner.hpp:107-113).
template <class S>
inline bool llt_inverse_upper(const S* a, int n, S* inv) {
  std::vector<S> L(n * n, S(0));
  for (int j = 0; j < n; ++j) {
    S d = a[j * n + j];  // (j,j)
    for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
    if (!(d > S(0))) return false;
    const S ljj = std::sqrt(d);
    L[j * n + j] = ljj;
    for (int i = j + 1; i < n; ++i) {
      S v = a[j * n + i];  // symmetric: (i,j) := upper (j,i)
      for (int k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = v / ljj;
    }
  }
  // solve L L^T X = I column by column
  std::vector<S> y(n);
  for (int c = 0; c < n; ++c) {
    for (int i = 0; i < n; ++i) {
      S v = (i == c) ? S(1) : S(0);
      for (int k = 0; k < i; ++k) v -= L[i * n + k] * y[k];
      y[i] = v / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      S v = y[i];
      for (int k = i + 1; k < n; ++k) v -= L[k * n + i] * inv[k * n + c];
      inv[i * n + c] = v / L[i * n + i];
    }
  }
  return true;
}

// Eigen Matrix3::inverse() (cofactor formula), used by the SC cross-check
// (src/rootba/sc/landmark_block.hpp:247, 440).
template <class S>
inline void inverse3(const S* m, S* o) {
  const S c00 = m[4] * m[8] - m[5] * m[7];
  const S c01 = m[5] * m[6] - m[3] * m[8];
  const S c02 = m[3] * m[7] - m[4] * m[6];
  const S det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  const S id = S(1) / det;
  o[0] = c00 * id;
  o[1] = (m[2] * m[7] - m[1] * m[8]) * id;
  o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c01 * id;
  o[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c02 * id;
  o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

template <class S>
inline bool all_finite(const S* v, int n) {
  for (int i = 0; i < n; ++i)
    if (!std::isfinite(v[i])) return false;
  return true;
}

// Scalar-typed dot / norm as Eigen evaluates them for VecX<Scalar>, widened to
// double by the caller (conjugate_gradient.hpp). Eight partial
// sums stand in for Eigen's packet reduction.
template <class S>
inline S dot(const S* a, const S* b, size_t n) {
  S acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  size_t i = 0;
  for (; i + 8 <= n; i += 8)
    for (int j = 0; j < 8; ++j) acc[j] += a[i + j] * b[i + j];
  S r = ((acc[0] + acc[4]) + (acc[1] + acc[5])) +
        ((acc[2] + acc[6]) + (acc[3] + acc[7]));
  for (; i < n; ++i) r += a[i] * b[i];
  return r;
}

// ---------------------------------------------------------------------------
// The solver state: BalProblem (row A) + LinearizationQR + LinearizorQR
// ---------------------------------------------------------------------------
template <class S>
class Oracle {
 public:
  static constexpr int P = 9;  // POSE_SIZE (linearizor_qr.hpp:51)

  // LinearizationQR ctor (src/rootba/qr/linearization_qr.hpp:80-111) +
  // LandmarkBlockDynamic::allocate_landmark_impl
  // (src/rootba/qr/landmark_block_dynamic.hpp:49-69).
  // Topology is CSR landmark -> observations, camera indices ascending inside
  // one landmark (std::map iteration order, bal_problem.hpp:131).
  Oracle(int n_cams, int n_lms, const int64_t* lm_obs_offsets,
         const int32_t* obs_cam_idx, const S* obs_xy, const Options& opt)
      : n_cams_(n_cams), n_lms_(n_lms), opt_(opt) {
    lm_off_.assign(lm_obs_offsets, lm_obs_offsets + n_lms + 1);
    n_obs_ = lm_off_[n_lms];
    obs_cam_.assign(obs_cam_idx, obs_cam_idx + n_obs_);
    obs_xy_.assign(obs_xy, obs_xy + 2 * n_obs_);
    cams_.assign(size_t(10) * n_cams, S(0));
    lms_.assign(size_t(3) * n_lms, S(0));
    cams_bak_ = cams_;
    lms_bak_ = lms_;
    blk_off_.resize(n_lms + 1);
    size_t off = 0;
    for (int l = 0; l < n_lms; ++l) {
      blk_off_[l] = off;
      // (solver_type 2, the matrix-free Schur-complement referee: no dense landmark blocks - that is its point)
      if (opt_.solver_type != 2) off += size_t(rows(l)) * cols(l);
    }
    blk_off_[n_lms] = off;
#ifdef _OPENMP
    n_threads_ = opt_.num_threads > 0 ? opt_.num_threads : omp_get_max_threads();
#else
    n_threads_ = 1;
#endif
    // Block storage is FIRST TOUCHED by the thread that will work on it: every landmark loop below is
    // an `omp for schedule(static)` over the same range with the same thread count, so a block stays
    // on the NUMA node of its worker (a zero-filling std::vector would put all of it on the
    // constructing thread's node - on a two-socket host every product then crosses the socket link).
    storage_size_ = off;
    storage_.reset(static_cast<S*>(std::aligned_alloc(64, (std::max<size_t>(off, 8) * sizeof(S) + 63) / 64 * 64)));
    if (!storage_) throw std::bad_alloc();
#pragma omp parallel for num_threads(n_threads_) schedule(static)
    for (int l = 0; l < n_lms; ++l)
      std::fill(storage_.get() + blk_off_[l], storage_.get() + blk_off_[l + 1], S(0));
    // per-thread accumulators of the camera-sized reductions, allocated (and first touched) once
    acc_.resize(size_t(n_threads_));
    xr_.resize(size_t(n_threads_));
    tmp_.resize(size_t(n_threads_));
#pragma omp parallel num_threads(n_threads_)
    {
#ifdef _OPENMP
      const int tid = omp_get_thread_num();
#else
      const int tid = 0;
#endif
      acc_[tid].assign(size_t(P) * n_cams, S(0));
    }
    jl_col_scale_.assign(size_t(3) * n_lms, S(0));
    rot_.assign(size_t(12) * n_lms, S(0));
    damped_.assign(n_lms, 0);
    failed_.assign(n_lms, 0);
    eps_ = opt_.jacobi_scaling_eps > 0 ? S(opt_.jacobi_scaling_eps)
                                       : epsilon_sqrt<S>();
  }

  int n_cams() const { return n_cams_; }
  int n_lms() const { return n_lms_; }
  int64_t n_obs() const { return n_obs_; }
  int n_threads() const { return n_threads_; }
  int k(int l) const { return int(lm_off_[l + 1] - lm_off_[l]); }
  int pad(int l) const { return (4 - (P * k(l)) % 4) % 4; }
  int lm_idx(int l) const { return P * k(l) + pad(l); }
  int res_idx(int l) const { return lm_idx(l) + 3; }
  int cols(int l) const { return lm_idx(l) + 4; }
  int rows(int l) const { return 2 * k(l) + 3; }
  S* block(int l) { return storage_.get() + blk_off_[l]; }
  const S* block(int l) const { return storage_.get() + blk_off_[l]; }
  std::vector<S>& cams() { return cams_; }
  std::vector<S>& lms() { return lms_; }
  const std::vector<S>& jl_col_scale() const { return jl_col_scale_; }
  const std::vector<S>& pose_jacobian_scaling() const { return pose_scaling_; }
  const std::vector<S>& precond_blocks() const { return precond_blocks_; }
  S eps() const { return eps_; }
  Options& options() { return opt_; }

  // BalProblem::backup / restore (src/rootba/bal/bal_problem.cpp:590-608)
  void backup() {
    cams_bak_ = cams_;
    lms_bak_ = lms_;
  }
  void restore() {
    cams_ = cams_bak_;
    lms_ = lms_bak_;
  }

  // BalBundleAdjustmentHelper::compute_error (helper.cpp:68-109) +
  // ResidualInfoAccu::add (src/rootba/bal/residual_info.cpp:97-110).
  void compute_error(ResidualInfo& out) const {
    const bool ignore_validity_check = !opt_.use_valid_projections_only;
    ResidualInfo acc;
    int nv = 1;
    long long n_all = 0, n_valid = 0;
    double e_all = 0, e_valid = 0, r_all = 0, r_valid = 0;
#pragma omp parallel for num_threads(n_threads_) schedule(static) \
    reduction(+ : n_all, n_valid, e_all, e_valid, r_all, r_valid) reduction(&& : nv)
    for (int l = 0; l < n_lms_; ++l) {
      for (int64_t o = lm_off_[l]; o < lm_off_[l + 1]; ++o) {
        S res[2];
        const bool projection_valid = linearize_point<S>(
            &obs_xy_[2 * o], &lms_[3 * l], &cams_[10 * obs_cam_[o]],
            ignore_validity_check, res, nullptr, nullptr, nullptr);
        const bool numerically_valid =
            std::isfinite(res[0]) && std::isfinite(res[1]);
        const S res_squared = res[0] * res[0] + res[1] * res[1];
        S werr, w;
        compute_error_weight<S>(opt_, res_squared, werr, w);
        nv = nv && numerically_valid;
        ++n_all;
        e_all += double(werr);
        r_all += double(std::sqrt(res_squared));
        if (projection_valid) {
          ++n_valid;
          e_valid += double(werr);
          r_valid += double(std::sqrt(res_squared));
        }
      }
    }
    acc.all.num_obs = int(n_all);
    acc.all.error = e_all;
    acc.all.residual_sum = r_all;
    acc.valid.num_obs = int(n_valid);
    acc.valid.error = e_valid;
    acc.valid.residual_sum = r_valid;
    acc.is_numerically_valid = nv;
    out = acc;
  }

  // -------------------------------------------------------------------------
  // Landmark block kernels (src/rootba/qr/impl/landmark_block_base.ipp)
  // -------------------------------------------------------------------------

  // linearize_landmark (ipp:88-147)
  void linearize_landmark(int l) {
    const int K = k(l), C = cols(l), R_ = rows(l);
    S* st = block(l);
    std::fill(st, st + size_t(R_) * C, S(0));
    damped_[l] = 0;
    bool numerically_valid = true;
    const int li = lm_idx(l), ri = res_idx(l);
    for (int i = 0; i < K; ++i) {
      const int64_t o = lm_off_[l] + i;
      S Jp[12], Ji[6], Jl[6], res[2];
      const bool valid =
          linearize_point<S>(&obs_xy_[2 * o], &lms_[3 * l],
                             &cams_[10 * obs_cam_[o]], true, res, Jp, Ji, Jl);
      if (!opt_.use_valid_projections_only || valid) {
        numerically_valid = numerically_valid && all_finite(Jl, 6) &&
                            all_finite(Jp, 12) && all_finite(Ji, 6) &&
                            all_finite(res, 2);
        const S res_squared = res[0] * res[0] + res[1] * res[1];
        S werr, w;
        compute_error_weight<S>(opt_, res_squared, werr, w);
        const S sw = std::sqrt(w);
        for (int r = 0; r < 2; ++r) {
          S* row = st + size_t(2 * i + r) * C;
          for (int c = 0; c < 6; ++c) row[P * i + c] = sw * Jp[6 * r + c];
          for (int c = 0; c < 3; ++c) row[P * i + 6 + c] = sw * Ji[3 * r + c];
          for (int c = 0; c < 3; ++c) row[li + c] = sw * Jl[3 * r + c];
          row[ri] = sw * res[r];
        }
      }
    }
    failed_[l] = numerically_valid ? 0 : 1;
  }

  // add_Jp_diag2 (ipp:493-518): column squared norms over the 2k residual rows
  void add_Jp_diag2(int l, S* res) const {
    const int K = k(l), C = cols(l);
    const S* st = block(l);
    for (int i = 0; i < K; ++i) {
      const int cam = obs_cam_[lm_off_[l] + i];
      for (int c = 0; c < P; ++c) {
        S s2 = 0;
        for (int r = 0; r < 2 * K; ++r) {
          const S v = st[size_t(r) * C + P * i + c];
          s2 += v * v;
        }
        res[P * cam + c] += s2;
      }
    }
  }

  // add_Jp_T_Jp_blockdiag (ipp:554-569): JACOBI preconditioner blocks
  void add_Jp_T_Jp_blockdiag(int l, S* blocks) const {
    const int K = k(l), C = cols(l);
    const S* st = block(l);
    for (int i = 0; i < K; ++i) {
      const int cam = obs_cam_[lm_off_[l] + i];
      const S* r0 = st + size_t(2 * i) * C + P * i;
      const S* r1 = st + size_t(2 * i + 1) * C + P * i;
      S* B = blocks + size_t(81) * cam;
      for (int a = 0; a < P; ++a)
        for (int b = 0; b < P; ++b) B[a * P + b] += r0[a] * r0[b] + r1[a] * r1[b];
    }
  }

  // scale_Jl_cols (ipp:571-587)
  void scale_Jl_cols(int l) {
    const int K = k(l), C = cols(l), li = lm_idx(l);
    S* st = block(l);
    for (int c = 0; c < 3; ++c) {
      S s2 = 0;
      for (int r = 0; r < 2 * K; ++r) {
        const S v = st[size_t(r) * C + li + c];
        s2 += v * v;
      }
      const S sc = S(1) / (eps_ + std::sqrt(s2));
      jl_col_scale_[3 * l + c] = sc;
      for (int r = 0; r < 2 * K; ++r) st[size_t(r) * C + li + c] *= sc;
    }
  }

  // perform_qr_householder (ipp:717-743) with Eigen's makeHouseholder /
  // applyHouseholderOnTheLeft restated per SURVEY.md A.3.
  void perform_qr_householder(int l) {
    const int C = cols(l), R_ = rows(l), li = lm_idx(l);
    S* st = block(l);
    std::vector<S> ess(R_), tmp(C);
    for (int kk = 0; kk < 3; ++kk) {
      const int remaining_rows = R_ - kk - 3;
      // makeHouseholder on x = storage.col(li+kk).segment(kk, remaining_rows)
      const S c0 = st[size_t(kk) * C + li + kk];
      S tail_sq = 0;
      for (int r = 1; r < remaining_rows; ++r) {
        const S v = st[size_t(kk + r) * C + li + kk];
        tail_sq += v * v;
      }
      S tau, beta;
      if (remaining_rows == 1 ||
          tail_sq <= std::numeric_limits<S>::min()) {
        tau = 0;
        beta = c0;
        for (int r = 1; r < remaining_rows; ++r) ess[r] = 0;
      } else {
        beta = std::sqrt(c0 * c0 + tail_sq);
        if (c0 >= S(0)) beta = -beta;
        for (int r = 1; r < remaining_rows; ++r)
          ess[r] = st[size_t(kk + r) * C + li + kk] / (c0 - beta);
        tau = (beta - c0) / beta;
      }
      if (tau == S(0)) continue;
      // applyHouseholderOnTheLeft: M <- M - tau v (v^T M), v = [1; ess]
      for (int c = 0; c < C; ++c) {
        S t = st[size_t(kk) * C + c];
        for (int r = 1; r < remaining_rows; ++r)
          t += ess[r] * st[size_t(kk + r) * C + c];
        tmp[c] = t;
      }
      for (int c = 0; c < C; ++c) st[size_t(kk) * C + c] -= tau * tmp[c];
      for (int r = 1; r < remaining_rows; ++r) {
        const S f = tau * ess[r];
        S* row = st + size_t(kk + r) * C;
        for (int c = 0; c < C; ++c) row[c] -= f * tmp[c];
      }
    }
  }

  // perform_qr_givens (ipp:700-715)
  void perform_qr_givens(int l) {
    const int C = cols(l), R_ = rows(l), li = lm_idx(l);
    S* st = block(l);
    for (int n = 0; n < 3; ++n) {
      for (int m = R_ - 4; m > n; --m) {
        S c, s;
        make_givens(st[size_t(m - 1) * C + li + n], st[size_t(m) * C + li + n],
                    c, s);
        apply_rot_rows(st + size_t(m) * C, st + size_t(m - 1) * C, C, c, s);
      }
    }
  }

  void perform_qr(int l) {
    if (opt_.use_householder)
      perform_qr_householder(l);
    else
      perform_qr_givens(l);
  }

  // scale_Jp_cols (ipp:589-614)
  void scale_Jp_cols(int l, const S* jacobian_scaling) {
    const int K = k(l), C = cols(l);
    S* st = block(l);
    for (int i = 0; i < K; ++i) {
      const int cam = obs_cam_[lm_off_[l] + i];
      for (int c = 0; c < P; ++c) {
        const S sc = jacobian_scaling[P * cam + c];
        for (int r = 0; r < 2 * K; ++r) st[size_t(r) * C + P * i + c] *= sc;
      }
    }
  }

  // set_landmark_damping (ipp:165-210)
  void set_landmark_damping(int l, S lambda) {
    const int C = cols(l), R_ = rows(l), li = lm_idx(l);
    S* st = block(l);
    S* rot = &rot_[size_t(12) * l];
    if (damped_[l]) {
      // undo dampening (ipp:175-186): adjoint() = (c, -s), reverse order
      int idx = 5;
      for (int n = 2; n >= 0; --n) {
        for (int m = n; m >= 0; --m) {
          const S c = rot[2 * idx], s = rot[2 * idx + 1];
          apply_rot_rows(st + size_t(R_ - 3 + n - m) * C, st + size_t(n) * C, C,
                         c, -s);
          --idx;
        }
      }
      damped_[l] = 0;
    }
    if (lambda == S(0)) {
      for (int d = 0; d < 3; ++d) st[size_t(R_ - 3 + d) * C + li + d] = 0;
    } else {
      const S sl = std::sqrt(lambda);
      for (int d = 0; d < 3; ++d) st[size_t(R_ - 3 + d) * C + li + d] = sl;
      int idx = 0;
      for (int n = 0; n < 3; ++n) {
        for (int m = 0; m <= n; ++m) {
          S c, s;
          make_givens(st[size_t(n) * C + li + n],
                      st[size_t(R_ - 3 + n - m) * C + li + n], c, s);
          rot[2 * idx] = c;
          rot[2 * idx + 1] = s;
          apply_rot_rows(st + size_t(R_ - 3 + n - m) * C, st + size_t(n) * C, C,
                         c, s);
          ++idx;
        }
      }
      damped_[l] = 1;
    }
  }

  // add_Q2TJp_T_Q2TJp_blockdiag (ipp:520-552): rows 3.. (incl. damping rows)
  void add_Q2TJp_T_Q2TJp_blockdiag(int l, S* blocks) const {
    const int K = k(l), C = cols(l), R_ = rows(l);
    const S* st = block(l);
    for (int i = 0; i < K; ++i) {
      const int cam = obs_cam_[lm_off_[l] + i];
      S* B = blocks + size_t(81) * cam;
      for (int r = 3; r < R_; ++r) {
        const S* row = st + size_t(r) * C + P * i;
        for (int a = 0; a < P; ++a)
          for (int b = 0; b < P; ++b) B[a * P + b] += row[a] * row[b];
      }
    }
  }

  // add_Q2TJp_T_Q2Tr (ipp:443-466)
  void add_Q2TJp_T_Q2Tr(int l, S* res) const {
    const int K = k(l), C = cols(l), R_ = rows(l), ri = res_idx(l);
    const S* st = block(l);
    for (int i = 0; i < K; ++i) {
      const int cam = obs_cam_[lm_off_[l] + i];
      for (int c = 0; c < P; ++c) {
        S acc = 0;
        for (int r = 3; r < R_; ++r)
          acc += st[size_t(r) * C + P * i + c] * st[size_t(r) * C + ri];
        res[P * cam + c] += acc;
      }
    }
  }

  // add_Q2TJp_T_Q2TJp_mult_x (ipp:400-441)
  void add_Q2TJp_T_Q2TJp_mult_x(int l, S* res, const S* x_pose,
                                std::vector<S>& xr, std::vector<S>& tmp) const {
    const int K = k(l), C = cols(l), R_ = rows(l);
    const S* st = block(l);
    xr.resize(P * K);
    tmp.resize(R_ - 3);
    for (int i = 0; i < K; ++i) {
      const int cam = obs_cam_[lm_off_[l] + i];
      for (int c = 0; c < P; ++c) xr[P * i + c] = x_pose[P * cam + c];
    }
    for (int r = 3; r < R_; ++r) {
      const S* row = st + size_t(r) * C;
      S acc = 0;
      for (int c = 0; c < P * K; ++c) acc += row[c] * xr[c];
      tmp[r - 3] = acc;
    }
    for (int c = 0; c < P * K; ++c) xr[c] = 0;
    for (int r = 3; r < R_; ++r) {
      const S* row = st + size_t(r) * C;
      const S t = tmp[r - 3];
      for (int c = 0; c < P * K; ++c) xr[c] += row[c] * t;
    }
    for (int i = 0; i < K; ++i) {
      const int cam = obs_cam_[lm_off_[l] + i];
      for (int c = 0; c < P; ++c) res[P * cam + c] += xr[P * i + c];
    }
  }

  // back_substitute (ipp:212-284). Returns false on non-finite increment
  // (the reference LOG(FATAL)s, ipp:266-279).
  bool back_substitute(int l, const S* pose_inc, double& l_diff) {
    const int K = k(l), C = cols(l), R_ = rows(l), li = lm_idx(l),
              ri = res_idx(l);
    S* st = block(l);
    std::vector<S> xr(P * K);
    for (int i = 0; i < K; ++i) {
      const int cam = obs_cam_[lm_off_[l] + i];
      for (int c = 0; c < P; ++c) xr[P * i + c] = pose_inc[P * cam + c];
    }
    // inc = -R^{-1} (Q1^T r + Q1^T Jp x), R upper triangular, damped
    S rhs[3];
    for (int r = 0; r < 3; ++r) {
      S acc = st[size_t(r) * C + ri];
      for (int c = 0; c < P * K; ++c) acc += st[size_t(r) * C + c] * xr[c];
      rhs[r] = acc;
    }
    S inc[3];
    for (int r = 2; r >= 0; --r) {
      S v = rhs[r];
      for (int c = r + 1; c < 3; ++c) v -= st[size_t(r) * C + li + c] * inc[c];
      inc[r] = v / st[size_t(r) * C + li + r];
    }
    for (int r = 0; r < 3; ++r) inc[r] = -inc[r];

    // undo damping before computing the model cost difference (ipp:247-248)
    set_landmark_damping(l, S(0));

    std::vector<S> g(R_ - 3);
    for (int r = 0; r < R_ - 3; ++r) {
      S acc = 0;
      for (int c = 0; c < P * K; ++c) acc += st[size_t(r) * C + c] * xr[c];
      g[r] = acc;
    }
    for (int r = 0; r < 3; ++r)
      for (int c = r; c < 3; ++c) g[r] += st[size_t(r) * C + li + c] * inc[c];
    S acc = 0;
    for (int r = 0; r < R_ - 3; ++r)
      acc += g[r] * (S(0.5) * g[r] + st[size_t(r) * C + ri]);
    l_diff -= double(acc);

    const bool ok = all_finite(inc, 3) && all_finite(&lms_[3 * l], 3);
    for (int c = 0; c < 3; ++c) lms_[3 * l + c] += inc[c] * jl_col_scale_[3 * l + c];
    return ok;
  }

  // -------------------------------------------------------------------------
  // LinearizationQR (src/rootba/qr/linearization_qr.hpp)
  // -------------------------------------------------------------------------

  // get_stage1 (linearization_qr.hpp:634-712). Returns false on numerical
  // failure (reference: empty vector). jacobi_blocks (81 n_c) optional.
  bool get_stage1(std::vector<S>& jp_diag2, std::vector<S>* jacobi_blocks) {
    const size_t n = size_t(P) * n_cams_;
    std::vector<std::vector<S>> acc(n_threads_), accb(n_threads_);
    int valid = 1;
#pragma omp parallel num_threads(n_threads_) reduction(&& : valid)
    {
#ifdef _OPENMP
      const int tid = omp_get_thread_num();
#else
      const int tid = 0;
#endif
      acc[tid].assign(n, S(0));
      if (jacobi_blocks) accb[tid].assign(size_t(81) * n_cams_, S(0));
#pragma omp for schedule(static)
      for (int l = 0; l < n_lms_; ++l) {
        linearize_landmark(l);
        if (!failed_[l]) {
          if (jacobi_blocks) add_Jp_T_Jp_blockdiag(l, accb[tid].data());
          add_Jp_diag2(l, acc[tid].data());
          scale_Jl_cols(l);
          perform_qr(l);
        } else {
          valid = 0;
        }
      }
    }
    jp_diag2.assign(n, S(0));
    for (int t = 0; t < n_threads_; ++t)
      for (size_t i = 0; i < n; ++i) jp_diag2[i] += acc[t][i];
    if (jacobi_blocks) {
      jacobi_blocks->assign(size_t(81) * n_cams_, S(0));
      for (int t = 0; t < n_threads_; ++t)
        for (size_t i = 0; i < jacobi_blocks->size(); ++i)
          (*jacobi_blocks)[i] += accb[t][i];
    }
    return valid != 0;
  }

  // set_pose_damping (linearization_qr.hpp:138-143)
  void set_pose_damping(S lambda) { pose_damping_ = lambda; }
  S pose_damping() const { return pose_damping_; }

  // get_stage2 (linearization_qr.hpp:716-815) + LandmarkBlockBase::stage2
  // (ipp:638-658).
  void get_stage2(S lambda, const S* jacobian_scaling,
                  std::vector<S>* precond_blocks, std::vector<S>& bref) {
    const size_t n = size_t(P) * n_cams_;
    std::vector<std::vector<S>> acc(n_threads_), accb(n_threads_);
#pragma omp parallel num_threads(n_threads_)
    {
#ifdef _OPENMP
      const int tid = omp_get_thread_num();
#else
      const int tid = 0;
#endif
      acc[tid].assign(n, S(0));
      if (precond_blocks) accb[tid].assign(size_t(81) * n_cams_, S(0));
#pragma omp for schedule(static)
      for (int l = 0; l < n_lms_; ++l) {
        if (jacobian_scaling) scale_Jp_cols(l, jacobian_scaling);
        set_landmark_damping(l, lambda);
        if (precond_blocks) add_Q2TJp_T_Q2TJp_blockdiag(l, accb[tid].data());
        add_Q2TJp_T_Q2Tr(l, acc[tid].data());
      }
    }
    bref.assign(n, S(0));
    for (int t = 0; t < n_threads_; ++t)
      for (size_t i = 0; i < n; ++i) bref[i] += acc[t][i];
    if (precond_blocks) {
      precond_blocks->assign(size_t(81) * n_cams_, S(0));
      for (int t = 0; t < n_threads_; ++t)
        for (size_t i = 0; i < precond_blocks->size(); ++i)
          (*precond_blocks)[i] += accb[t][i];
      // add pose damping to the preconditioner (linearization_qr.hpp:796-802)
      if (pose_damping_ > 0)
        for (int c = 0; c < n_cams_; ++c)
          for (int d = 0; d < P; ++d)
            (*precond_blocks)[size_t(81) * c + d * P + d] += pose_damping_;
    }
  }

  // right_multiply == get_Q2TJp_T_Q2TJp_mult_x
  // (linearization_qr.hpp:406-429, 821-825)
  void right_multiply(const S* x, S* y) const {
    const size_t n = size_t(P) * n_cams_;
    if (opt_.solver_type == 2) {
      // the same system without its matrix: (S + lambda I) x = (Hpp + lambda I) x - E0 x from the per-observation
      // Jacobians (sc_mf_prepare)
      power_e0_mult(x, y);
#pragma omp parallel for num_threads(n_threads_) schedule(static)
      for (int c = 0; c < n_cams_; ++c) {
        const S* B = &mf_Hpp_[size_t(81) * c];
        const S* xc = x + size_t(P) * c;
        for (int a = 0; a < P; ++a) {
          S v = 0;
          for (int bb = 0; bb < P; ++bb) v += B[a * P + bb] * xc[bb];
          y[size_t(P) * c + a] = v - y[size_t(P) * c + a];
        }
      }
      return;
    }
    if (opt_.solver_type == 1) {
      // BlockSparseMatrix::right_multiply on H_pp (block_sparse_matrix.hpp); H already
      // holds the pose damping (linearization_sc.hpp:323-327)
#pragma omp parallel for num_threads(n_threads_) schedule(static)
      for (int64_t i = 0; i < int64_t(n); ++i) {
        S v = 0;
        const S* row = &sc_H_[size_t(i) * n];
        for (size_t j = 0; j < n; ++j) v += row[j] * x[j];
        y[i] = v;
      }
      return;
    }
    // reduction_alg = 0 flavour of the reference (per-thread accumulators, linearization_qr.hpp:294-334),
    // static schedule = the first-touch partition of the block storage, parallel combine
#pragma omp parallel num_threads(n_threads_)
    {
#ifdef _OPENMP
      const int tid = omp_get_thread_num();
#else
      const int tid = 0;
#endif
      std::fill(acc_[tid].begin(), acc_[tid].end(), S(0));
      std::vector<S>& xr = xr_scratch(tid);
      std::vector<S>& tmp = tmp_scratch(tid);
#pragma omp for schedule(static)
      for (int l = 0; l < n_lms_; ++l)
        add_Q2TJp_T_Q2TJp_mult_x(l, acc_[tid].data(), x, xr, tmp);
#pragma omp for schedule(static)
      for (int64_t i = 0; i < int64_t(n); ++i) {
        S v = 0;
        for (int t = 0; t < n_threads_; ++t) v += acc_[t][i];
        y[i] = v + (pose_damping_ > 0 ? x[i] * pose_damping_ : S(0));
      }
    }
  }

  // LinearizationQR::back_substitute (linearization_qr.hpp:165-179)
  S back_substitute_all(const S* pose_inc, bool* finite_ok = nullptr) {
    double l_diff = 0;
    int ok = 1;
#pragma omp parallel for num_threads(n_threads_) schedule(static) \
    reduction(+ : l_diff) reduction(&& : ok)
    for (int l = 0; l < n_lms_; ++l) {
      double ld = 0;
      const bool o = back_substitute(l, pose_inc, ld);
      l_diff += ld;
      ok = ok && o;
    }
    if (finite_ok) *finite_ok = ok != 0;
    return S(l_diff);
  }

  // -------------------------------------------------------------------------
  // PCG (src/rootba/cg/conjugate_gradient.hpp:113-298) with
  // BlockDiagonalPreconditioner (src/rootba/cg/preconditioner.hpp:79-136)
  // -------------------------------------------------------------------------
  bool build_preconditioner(const std::vector<S>& blocks, const S* diagonal) {
    inv_blocks_.assign(size_t(81) * n_cams_, S(0));
    int ok = 1;
#pragma omp parallel for num_threads(n_threads_) schedule(static) reduction(&& : ok)
    for (int c = 0; c < n_cams_; ++c) {
      S tmp[81];
      for (int i = 0; i < 81; ++i) tmp[i] = blocks[size_t(81) * c + i];
      if (diagonal)
        for (int d = 0; d < P; ++d) tmp[d * P + d] += diagonal[P * c + d];
      ok = ok && llt_inverse_upper<S>(tmp, P, &inv_blocks_[size_t(81) * c]);
    }
    return ok != 0;
  }

  void precond_solve(const S* b, S* x) const {
    for (int c = 0; c < n_cams_; ++c) {
      const S* M = &inv_blocks_[size_t(81) * c];
      for (int i = 0; i < P; ++i) {
        S acc = 0;
        for (int j = 0; j < P; ++j) acc += M[i * P + j] * b[P * c + j];
        x[P * c + i] = acc;
      }
    }
  }

  // -------------------------------------------------------------------------
  // PowerSCPreconditioner (src/rootba/cg/preconditioner.hpp:145-254) on the
  // landmark pieces of LandmarkBlockSC (src/rootba/sc/landmark_block.hpp:
  // 342-364 stage/Hll_inv, 381-407 add_Jp_x/add_JpT_x, get_jacobi
  // linearization_sc.hpp:244-264):
  //   x = sum_{i=0..order} (Hpp^-1 E0)^i Hpp^-1 b,
  //   E0 v = sum_l Jp^T Jl Hll^-1 Jl^T Jp v,  Hll^-1 = (Jl^T Jl + lambda I)^-1,
  //   Hpp = sum Jp^T Jp + lambda I  (scaled Jacobians, JACOBI blocks).
  // The reference wires it only for the SC solver (linearizor_sc.cpp:166-171);
  // with the QR solver it is a new combination whose operator
  // is mathematically the same, so this is its oracle.
  // -------------------------------------------------------------------------
  void power_precond_prepare(S lambda) {
    pw_Jp_.assign(size_t(18) * n_obs_, S(0));
    pw_Jl_.assign(size_t(6) * n_obs_, S(0));
    pw_Hll_inv_.assign(size_t(9) * n_lms_, S(0));
    std::vector<S> Hpp(size_t(81) * n_cams_, S(0));
    for (int l = 0; l < n_lms_; ++l) {
      const int K = k(l);
      std::vector<S> Jp(size_t(2 * K) * P), Jl(size_t(2 * K) * 3), r(2 * K);
      sc_linearize(l, pose_scaling_.data(), Jp, Jl, r, nullptr, nullptr);
      S Hll[9] = {0};
      for (int row = 0; row < 2 * K; ++row)
        for (int a = 0; a < 3; ++a)
          for (int c = 0; c < 3; ++c) Hll[a * 3 + c] += Jl[row * 3 + a] * Jl[row * 3 + c];
      for (int d = 0; d < 3; ++d) Hll[d * 3 + d] += lambda;
      inverse3(Hll, &pw_Hll_inv_[size_t(9) * l]);
      const int64_t o0 = lm_off_[l];
      std::copy(Jp.begin(), Jp.end(), pw_Jp_.begin() + 18 * o0);
      std::copy(Jl.begin(), Jl.end(), pw_Jl_.begin() + 6 * o0);
      for (int i = 0; i < K; ++i) {
        S* B = &Hpp[size_t(81) * obs_cam_[o0 + i]];
        for (int a = 0; a < P; ++a)
          for (int bb = 0; bb < P; ++bb)
            B[a * P + bb] += Jp[(2 * i) * P + a] * Jp[(2 * i) * P + bb] +
                             Jp[(2 * i + 1) * P + a] * Jp[(2 * i + 1) * P + bb];
      }
    }
    std::vector<S> diag(size_t(P) * n_cams_, lambda);
    build_preconditioner(Hpp, diag.data());  // inv_blocks_ = Hpp^-1
  }

  // right_mul_e0 (preconditioner.hpp:223-245); landmarks in parallel into the per-thread accumulators of the products
  // (static schedule, combined in thread order: the result does not depend on timing)
  void power_e0_mult(const S* x, S* res) const {
    const size_t n = size_t(P) * n_cams_;
    // (the power-series preconditioner of solver types 0 / 1 keeps the serial landmark order its golden vectors were
    //  made with; the matrix-free referee, solver_type 2, runs this product on all threads)
    const int nt = opt_.solver_type == 2 ? n_threads_ : 1;
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
      const int tid = omp_get_thread_num();
#else
      const int tid = 0;
#endif
      S* acc = acc_[tid].data();
      std::fill(acc, acc + n, S(0));
      std::vector<S> Jp_x;
#pragma omp for schedule(static)
      for (int l = 0; l < n_lms_; ++l) {
        const int K = k(l);
        const int64_t o0 = lm_off_[l];
        const S* Jp = &pw_Jp_[18 * o0];
        const S* Jl = &pw_Jl_[6 * o0];
        const S* Hi = &pw_Hll_inv_[size_t(9) * l];
        Jp_x.assign(2 * K, S(0));
        S JlT[3] = {0, 0, 0};
        for (int i = 0; i < K; ++i) {
          const S* v = x + P * obs_cam_[o0 + i];
          for (int rr = 0; rr < 2; ++rr) {
            S a2 = 0;
            for (int a = 0; a < P; ++a) a2 += Jp[(2 * i + rr) * P + a] * v[a];
            Jp_x[2 * i + rr] = a2;
            for (int c = 0; c < 3; ++c) JlT[c] += Jl[(2 * i + rr) * 3 + c] * a2;
          }
        }
        S h[3];
        for (int a = 0; a < 3; ++a) h[a] = Hi[a * 3] * JlT[0] + Hi[a * 3 + 1] * JlT[1] + Hi[a * 3 + 2] * JlT[2];
        for (int i = 0; i < K; ++i) {
          S* out = acc + P * obs_cam_[o0 + i];
          for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * i + rr;
            const S t = Jl[row * 3] * h[0] + Jl[row * 3 + 1] * h[1] + Jl[row * 3 + 2] * h[2];
            for (int a = 0; a < P; ++a) out[a] += Jp[row * P + a] * t;
          }
        }
      }
#pragma omp for schedule(static)
      for (int64_t i = 0; i < int64_t(n); ++i) {
        S v = 0;
        for (int t = 0; t < nt; ++t) v += acc_[t][i];
        res[i] = v;
      }
    }
  }

  // -------------------------------------------------------------------------
  // solver_type 2: the Schur-complement system of solver_type 1 (LinearizorSC, linearizor_sc.cpp:101-186) WITHOUT its
  // matrix - the dense H_pp of sc_build is 9 n_c x 9 n_c scalars (121 GB for final-13682 in double) and the dense
  // landmark blocks of the square-root solver 55 GB, while the per-observation Jacobians are 24 scalars per observation
  // (5.6 GB). (S + lambda I) x = (Hpp + lambda I) x - E0 x with E0 as in the power series above; SCHUR_JACOBI
  // preconditioner = the inverted diagonal blocks of S + lambda I. Same system, same PCG recurrence: in exact arithmetic
  // the iterates are those of solver types 0 and 1 - in float64 this is the INDEPENDENT referee of the float32 runs at
  // sizes where the other two do not fit the host (tests/test_gpu_baseline_configs.py).
  // -------------------------------------------------------------------------
  void sc_mf_prepare(S lambda) {
    pw_Jp_.assign(size_t(18) * n_obs_, S(0));
    pw_Jl_.assign(size_t(6) * n_obs_, S(0));
    pw_Hll_inv_.assign(size_t(9) * n_lms_, S(0));
    mf_Hpp_.assign(size_t(81) * n_cams_, S(0));
    precond_blocks_.assign(size_t(81) * n_cams_, S(0));
    std::vector<S> Jp, Jl, r;
    for (int l = 0; l < n_lms_; ++l) {
      const int K = k(l);
      Jp.assign(size_t(2 * K) * P, S(0));
      Jl.assign(size_t(2 * K) * 3, S(0));
      r.assign(2 * K, S(0));
      sc_linearize(l, pose_scaling_.data(), Jp, Jl, r, nullptr, nullptr);
      S Hll[9] = {0}, Hinv[9];
      for (int row = 0; row < 2 * K; ++row)
        for (int a = 0; a < 3; ++a)
          for (int c = 0; c < 3; ++c) Hll[a * 3 + c] += Jl[row * 3 + a] * Jl[row * 3 + c];
      for (int d = 0; d < 3; ++d) Hll[d * 3 + d] += lambda;
      inverse3(Hll, Hinv);
      const int64_t o0 = lm_off_[l];
      std::copy(Hinv, Hinv + 9, pw_Hll_inv_.begin() + size_t(9) * l);
      std::copy(Jp.begin(), Jp.end(), pw_Jp_.begin() + 18 * o0);
      std::copy(Jl.begin(), Jl.end(), pw_Jl_.begin() + 6 * o0);
      for (int i = 0; i < K; ++i) {
        const int ci = obs_cam_[o0 + i];
        S W[27];  // W_i = Jp_i^T Jl_i (9 x 3)
        for (int a = 0; a < P; ++a)
          for (int c = 0; c < 3; ++c)
            W[a * 3 + c] = Jp[(2 * i) * P + a] * Jl[(2 * i) * 3 + c] + Jp[(2 * i + 1) * P + a] * Jl[(2 * i + 1) * 3 + c];
        S* G = &mf_Hpp_[size_t(81) * ci];
        S* D = &precond_blocks_[size_t(81) * ci];
        for (int a = 0; a < P; ++a) {
          S wa[3];
          for (int c = 0; c < 3; ++c) wa[c] = W[a * 3] * Hinv[c] + W[a * 3 + 1] * Hinv[3 + c] + W[a * 3 + 2] * Hinv[6 + c];
          for (int bb = 0; bb < P; ++bb) {
            const S g = Jp[(2 * i) * P + a] * Jp[(2 * i) * P + bb] + Jp[(2 * i + 1) * P + a] * Jp[(2 * i + 1) * P + bb];
            G[a * P + bb] += g;
            D[a * P + bb] += g - (wa[0] * W[bb * 3] + wa[1] * W[bb * 3 + 1] + wa[2] * W[bb * 3 + 2]);
          }
        }
      }
    }
    for (int c = 0; c < n_cams_; ++c)
      for (int a = 0; a < P; ++a) {
        mf_Hpp_[size_t(81) * c + a * P + a] += lambda;
        precond_blocks_[size_t(81) * c + a * P + a] += lambda;
      }
    build_preconditioner(precond_blocks_, nullptr);
  }

  // solve_assign (preconditioner.hpp:180-192)
  void power_precond_solve(const S* b, S* x) const {
    const size_t n = size_t(P) * n_cams_;
    std::vector<S> tmp(n), e(n);
    precond_solve(b, x);
    std::copy(x, x + n, tmp.begin());
    for (int i = 1; i <= opt_.power_order; ++i) {
      power_e0_mult(tmp.data(), e.data());
      precond_solve(e.data(), tmp.data());
      for (size_t j = 0; j < n; ++j) x[j] += tmp[j];
    }
  }

  CgSummary pcg(const std::vector<S>& bref, std::vector<S>& xref) {
    CgSummary summary;
    const size_t n = size_t(P) * n_cams_;
    const double min_it = opt_.min_cg_it;
    const int max_it = opt_.max_cg_it;
    const int residual_reset_period = 10;
    const double q_tolerance = opt_.eta;
    const double r_tolerance = -1.0;

    const double norm_b = std::sqrt(double(dot(bref.data(), bref.data(), n)));
    if (norm_b == 0.0) {
      std::fill(xref.begin(), xref.end(), S(0));
      summary.termination_type = 1;
      return summary;
    }
    std::vector<S> r(n), p(n), z(n), tmp(n), q(n), br(n);
    const double tol_r = r_tolerance * norm_b;
    right_multiply(xref.data(), tmp.data());
    for (size_t i = 0; i < n; ++i) r[i] = bref[i] - tmp[i];
    double norm_r = std::sqrt(double(dot(r.data(), r.data(), n)));
    if (min_it == 0 && norm_r <= tol_r) {
      summary.termination_type = 1;
      return summary;
    }
    double rho = 1.0;
    for (size_t i = 0; i < n; ++i) br[i] = bref[i] + r[i];
    double q0 = -1.0 * double(dot(xref.data(), br.data(), n));

    for (summary.num_iterations = 1;; ++summary.num_iterations) {
      if (opt_.preconditioner_type == 2)
        power_precond_solve(r.data(), z.data());
      else
        precond_solve(r.data(), z.data());
      const double last_rho = rho;
      rho = double(dot(r.data(), z.data(), n));
      if (rho == 0.0 || std::isinf(rho)) {
        summary.termination_type = 2;
        break;
      }
      if (summary.num_iterations == 1) {
        p = z;
      } else {
        const double beta = rho / last_rho;
        if (beta == 0.0 || std::isinf(beta)) {
          summary.termination_type = 2;
          break;
        }
        const S b = S(beta);
        for (size_t i = 0; i < n; ++i) p[i] = z[i] + b * p[i];
      }
      right_multiply(p.data(), q.data());
      const double pq = double(dot(p.data(), q.data(), n));
      if ((pq <= 0) || std::isinf(pq)) {
        summary.termination_type = 0;
        break;
      }
      const double alpha = rho / pq;
      if (std::isinf(alpha)) {
        summary.termination_type = 2;
        break;
      }
      const S a = S(alpha);
      for (size_t i = 0; i < n; ++i) xref[i] = xref[i] + a * p[i];
      if (summary.num_iterations % residual_reset_period == 0) {
        right_multiply(xref.data(), tmp.data());
        for (size_t i = 0; i < n; ++i) r[i] = bref[i] - tmp[i];
      } else {
        for (size_t i = 0; i < n; ++i) r[i] = r[i] - a * q[i];
      }
      for (size_t i = 0; i < n; ++i) br[i] = bref[i] + r[i];
      const double q1 = -1.0 * double(dot(xref.data(), br.data(), n));
      const double zeta = summary.num_iterations * (q1 - q0) / q1;
      if (zeta < q_tolerance && summary.num_iterations >= min_it) {
        summary.termination_type = 1;
        break;
      }
      q0 = q1;
      norm_r = std::sqrt(double(dot(r.data(), r.data(), n)));
      if (norm_r <= tol_r && summary.num_iterations >= min_it) {
        summary.termination_type = 1;
        break;
      }
      if (summary.num_iterations >= max_it) break;
    }
    return summary;
  }

  // -------------------------------------------------------------------------
  // LinearizorQR (src/rootba/solver/linearizor_qr.cpp:78-291), staged path
  // -------------------------------------------------------------------------

  // linearize() (linearizor_qr.cpp:78-138). Returns false on numerical failure
  // (reference CHECK-aborts, :121-122).
  bool linearize(LmIteration* it = nullptr) {
    const double t0 = now_seconds();
    if (opt_.solver_type >= 1) {
      // LinearizorSC::linearize (linearizor_sc.cpp:70-99): linearize_problem, get_Jp_diag2,
      // scale_Jl_cols, pose scaling from the UNSCALED Jp column norms
      const size_t n = size_t(P) * n_cams_;
      jp_diag2_.assign(n, S(0));
      bool ok = true;
      for (int l = 0; l < n_lms_; ++l) {
        const int K = k(l);
        std::vector<S> Jp(size_t(2 * K) * P), Jl(size_t(2 * K) * 3), r(2 * K);
        sc_linearize(l, nullptr, Jp, Jl, r, jp_diag2_.data(), nullptr);
        for (S v : Jp) ok = ok && std::isfinite(v);
        for (S v : Jl) ok = ok && std::isfinite(v);
        for (S v : r) ok = ok && std::isfinite(v);
      }
      if (!ok) return false;
      pose_scaling_.resize(n);
      for (size_t i = 0; i < n; ++i) pose_scaling_[i] = S(1) / (eps_ + std::sqrt(jp_diag2_[i]));
      new_linearization_point_ = true;
      if (it) it->stage1_time = now_seconds() - t0;
      return true;
    }
    const bool use_jacobi = opt_.preconditioner_type == 0;
    const bool ok =
        get_stage1(jp_diag2_, use_jacobi ? &precond_blocks_ : nullptr);
    if (!ok) return false;
    const size_t n = size_t(P) * n_cams_;
    pose_scaling_.resize(n);
    for (size_t i = 0; i < n; ++i)
      pose_scaling_[i] = S(1) / (eps_ + std::sqrt(jp_diag2_[i]));
    new_linearization_point_ = true;
    if (it) it->stage1_time = now_seconds() - t0;
    return true;
  }

  // solve(lambda) (linearizor_qr.cpp:141-265). Returns the (negated) scaled
  // pose increment, linearizor_base.cpp:100.
  std::vector<S> solve(S lambda, CgSummary* cg_out = nullptr,
                       LmIteration* it = nullptr) {
    double t0 = now_seconds();
    if (opt_.solver_type >= 1) {
      // LinearizorSC::solve (linearizor_sc.cpp:101-186): H_pp, b_p with pose + landmark
      // damping lambda, SCHUR_JACOBI = inverted diagonal blocks of H_pp, PCG
      const size_t n = size_t(P) * n_cams_;
      sc_lambda_ = lambda;
      std::vector<S> b;
      sc_build(lambda, lambda, pose_scaling_.data(), opt_.solver_type == 2 ? nullptr : &sc_H_, b, nullptr);
      b_ = b;
      if (it) it->stage2_time = now_seconds() - t0;
      t0 = now_seconds();
      if (opt_.solver_type == 2) {
        if (opt_.preconditioner_type != 1)
          throw std::invalid_argument("solver_type 2 (matrix-free Schur complement): SCHUR_JACOBI only");
        sc_mf_prepare(lambda);
      } else if (opt_.preconditioner_type == 2) {
        // POWER_SCHUR_COMPLEMENT (linearizor_sc.cpp:163-170): PowerSCPreconditioner on the JACOBI blocks
        // (get_jacobi) and the landmark blocks; pcg() applies it through power_precond_solve
        power_precond_prepare(lambda);
      } else {
        precond_blocks_.assign(size_t(81) * n_cams_, S(0));
        for (int c = 0; c < n_cams_; ++c)
          for (int a = 0; a < P; ++a)
            for (int bb = 0; bb < P; ++bb)
              precond_blocks_[size_t(81) * c + a * P + bb] = sc_H_[(size_t(P) * c + a) * n + P * c + bb];
        build_preconditioner(precond_blocks_, nullptr);
      }
      if (it) it->precond_time = now_seconds() - t0;
      t0 = now_seconds();
      std::vector<S> inc(n, S(0));
      CgSummary cg = pcg(b, inc);
      for (size_t i = 0; i < n; ++i) inc[i] = -inc[i];
      if (it) it->pcg_time = now_seconds() - t0;
      if (cg_out) *cg_out = cg;
      new_linearization_point_ = false;
      return inc;
    }
    const bool use_schur_jacobi = opt_.preconditioner_type == 1;
    set_pose_damping(lambda);
    std::vector<S> b;
    get_stage2(lambda,
               new_linearization_point_ ? pose_scaling_.data() : nullptr,
               use_schur_jacobi ? &precond_blocks_ : nullptr, b);
    b_ = b;
    if (it) it->stage2_time = now_seconds() - t0;
    t0 = now_seconds();
    const size_t n = size_t(P) * n_cams_;
    if (opt_.preconditioner_type == 0) {
      // scale_jacobians (block_sparse_matrix.hpp:89-100): D B D, once per
      // linearisation point (linearizor_qr.cpp:220-222)
      if (new_linearization_point_) {
        for (int c = 0; c < n_cams_; ++c)
          for (int a = 0; a < P; ++a)
            for (int bb = 0; bb < P; ++bb)
              precond_blocks_[size_t(81) * c + a * P + bb] *=
                  pose_scaling_[P * c + a] * pose_scaling_[P * c + bb];
      }
      std::vector<S> diag(n, lambda);
      build_preconditioner(precond_blocks_, diag.data());
    } else if (opt_.preconditioner_type == 2) {
      power_precond_prepare(lambda);
    } else {
      build_preconditioner(precond_blocks_, nullptr);
    }
    if (it) it->precond_time = now_seconds() - t0;
    t0 = now_seconds();
    std::vector<S> inc(n, S(0));
    CgSummary cg = pcg(b, inc);
    for (size_t i = 0; i < n; ++i) inc[i] = -inc[i];
    if (it) it->pcg_time = now_seconds() - t0;
    if (cg_out) *cg_out = cg;
    new_linearization_point_ = false;
    return inc;
  }

  // apply(inc) (linearizor_qr.cpp:268-291)
  S apply(std::vector<S> inc, LmIteration* it = nullptr) {
    const double t0 = now_seconds();
    bool ok = true;
    const S l_diff = opt_.solver_type >= 1
                         ? sc_back_substitute(sc_lambda_, pose_scaling_.data(), inc.data())
                         : back_substitute_all(inc.data(), &ok);
    if (it) it->backsub_time = now_seconds() - t0;
    if (!std::isfinite(l_diff) || !ok)
      return std::numeric_limits<S>::quiet_NaN();
    for (size_t i = 0; i < inc.size(); ++i) inc[i] *= pose_scaling_[i];
    for (int c = 0; c < n_cams_; ++c)
      apply_inc_camera<S>(&cams_[10 * c], &inc[P * c]);
    return l_diff;
  }

  const std::vector<S>& last_b() const { return b_; }
  const std::vector<S>& jp_diag2() const { return jp_diag2_; }

  // -------------------------------------------------------------------------
  // optimize_lm_ours (src/rootba/solver/bal_bundle_adjustment.cpp:249-544)
  // Returns number of log rows written (<= max_rows). termination: 0
  // NO_CONVERGENCE, 1 CONVERGENCE (function tolerance), -1 numerical failure.
  // -------------------------------------------------------------------------
  int optimize_lm(LmIteration* log, int max_rows, int* termination_out) {
    const S min_lambda = S(1.0 / opt_.max_trust_region_radius);
    const S max_lambda = S(1.0 / opt_.min_trust_region_radius);
    const S vee_factor = S(opt_.vee_factor);
    const S initial_vee = S(opt_.initial_vee);
    const int max_lm_iter = opt_.max_num_iterations;
    S lambda = S(1.0 / opt_.initial_trust_region_radius);
    S lambda_vee = initial_vee;
    int n_rows = 0;
    int termination = 0;
    bool terminated = false;
    double prev_cost_all = 0, prev_cost_valid = 0;

    auto push = [&](const LmIteration& r) {
      if (n_rows < max_rows) log[n_rows] = r;
      ++n_rows;
    };

    for (int it = 0; it <= max_lm_iter && !terminated;) {
      LmIteration row;
      row.iteration = it;
      double t_it = now_seconds();
      ResidualInfo ri;
      {
        const double t0 = now_seconds();
        compute_error(ri);
        row.residual_time += now_seconds() - t0;
      }
      if (!ri.is_numerically_valid) {
        termination = -1;
        break;
      }
      if (it == 0) {
        row.cost = ri.all.error;
        row.cost_valid = ri.valid.error;
        row.num_obs = ri.all.num_obs;
        row.num_obs_valid = ri.valid.num_obs;
        row.residual_sum = ri.all.residual_sum;
        row.residual_sum_valid = ri.valid.residual_sum;
        row.lambda = lambda;
        row.step_is_successful = 1;
        row.step_is_valid = 1;
        row.iteration_time = now_seconds() - t_it;
        push(row);
        prev_cost_all = ri.all.error;
        prev_cost_valid = ri.valid.error;
        ++it;
        continue;
      }
      if (!linearize(&row)) {
        termination = -1;
        break;
      }
      for (int j = 0; it <= max_lm_iter && !terminated; ++j) {
        if (j > 0) {
          row = LmIteration();
          row.iteration = it;
          t_it = now_seconds();
        }
        row.lambda = lambda;
        CgSummary cg;
        std::vector<S> inc = solve(lambda, &cg, &row);
        row.cg_iterations = cg.num_iterations;
        row.cg_termination = cg.termination_type;
        row.inc_norm =
            std::sqrt(double(dot(inc.data(), inc.data(), inc.size())));
        if (!all_finite(inc.data(), int(inc.size()))) {
          row.step_is_valid = 0;
          row.step_is_successful = 0;
          lambda = lambda_vee * lambda;
          lambda_vee *= vee_factor;
          row.iteration_time = now_seconds() - t_it;
          // the reference leaves it_summary.cost default-constructed (zeros)
          // on this path (bal_bundle_adjustment.cpp:360-399)
          row.cost = 0;
          row.cost_valid = 0;
          push(row);
          prev_cost_all = 0;
          prev_cost_valid = 0;
          ++it;
          if (lambda > max_lambda) terminated = true;
          continue;
        }
        backup();
        S l_diff = apply(inc, &row);
        ResidualInfo ri2;
        {
          const double t0 = now_seconds();
          compute_error(ri2);
          row.residual_time += now_seconds() - t0;
        }
        row.cost = ri2.all.error;
        row.cost_valid = ri2.valid.error;
        row.num_obs = ri2.all.num_obs;
        row.num_obs_valid = ri2.valid.num_obs;
        row.residual_sum = ri2.all.residual_sum;
        row.residual_sum_valid = ri2.valid.residual_sum;
        row.l_diff = l_diff;
        if (!std::isfinite(l_diff)) {
          row.step_is_valid = 0;
          row.step_is_successful = 0;
        } else if (!ri2.is_numerically_valid) {
          row.step_is_valid = 0;
          row.step_is_successful = 0;
        } else {
          S f_diff;
          if (opt_.optimized_cost == 0)
            f_diff = S(ri.all.error - ri2.all.error);
          else if (opt_.optimized_cost == 1)
            f_diff = S(ri.valid.error - ri2.valid.error);
          else
            f_diff = S(ri.valid.error_avg() - ri2.valid.error_avg());
          if (opt_.optimized_cost == 2) l_diff /= S(ri.valid.num_obs);
          const S step_quality = f_diff / l_diff;
          row.relative_decrease = step_quality;
          row.step_is_valid = l_diff > 0;
          row.step_is_successful =
              row.step_is_valid &&
              step_quality > S(opt_.min_relative_decrease);
        }
        if (row.step_is_successful) {
          lambda *= S(std::max(
              1.0 / 3, 1 - std::pow(2 * row.relative_decrease - 1, 3)));
          lambda = std::max(min_lambda, lambda);
          lambda_vee = initial_vee;
          row.iteration_time = now_seconds() - t_it;
          push(row);
          ++it;
          // function_tolerance_reached (bal_bundle_adjustment.cpp:181-207)
          double cost, change;
          if (opt_.optimized_cost == 0) {
            cost = ri2.all.error;
            change = std::abs(prev_cost_all - ri2.all.error);
          } else {
            cost = ri2.valid.error;
            change = std::abs(prev_cost_valid - ri2.valid.error);
          }
          prev_cost_all = ri2.all.error;
          prev_cost_valid = ri2.valid.error;
          if (change <= opt_.function_tolerance * cost) {
            terminated = true;
            termination = 1;
          }
          break;
        } else {
          lambda = lambda_vee * lambda;
          lambda_vee *= vee_factor;
          row.iteration_time = now_seconds() - t_it;
          push(row);
          // cost_change in the reference compares with the previous pushed
          // iteration (bal_bundle_adjustment.cpp:67-70); a rejected step
          // still becomes "previous" for the next row.
          prev_cost_all = ri2.all.error;
          prev_cost_valid = ri2.valid.error;
          restore();
          ++it;
          if (lambda > max_lambda) terminated = true;
        }
      }
    }
    if (termination_out) *termination_out = termination;
    return n_rows;
  }

  // -------------------------------------------------------------------------
  // Explicit Schur-complement cross-check: LandmarkBlockSC
  // (src/rootba/sc/landmark_block.hpp:127-279, 409-446) + LinearizationSC.
  // Dense H (9n_c x 9n_c, row-major) - small problems only. Uses the CURRENT
  // state, Jl column scaling with eps, Jp column scaling with `pose_scaling`
  // (may be null), landmark damping `lambda`, pose damping `pose_lambda`.
  // -------------------------------------------------------------------------
  void sc_build(S lambda, S pose_lambda, const S* pose_scaling,
                std::vector<S>* H, std::vector<S>& b,
                std::vector<S>* jp_diag2) const {
    const size_t n = size_t(P) * n_cams_;
    if (H) H->assign(n * n, S(0));
    b.assign(n, S(0));
    if (jp_diag2) jp_diag2->assign(n, S(0));
    for (int l = 0; l < n_lms_; ++l) {
      const int K = k(l);
      std::vector<S> Jp(size_t(2 * K) * P), Jl(size_t(2 * K) * 3), r(2 * K);
      sc_linearize(l, pose_scaling, Jp, Jl, r, jp_diag2 ? jp_diag2->data() : nullptr, nullptr);
      S Hll[9] = {0}, Hinv[9], Jltr[3] = {0};
      for (int rr = 0; rr < 2 * K; ++rr)
        for (int a = 0; a < 3; ++a) {
          Jltr[a] += Jl[rr * 3 + a] * r[rr];
          for (int c = 0; c < 3; ++c) Hll[a * 3 + c] += Jl[rr * 3 + a] * Jl[rr * 3 + c];
        }
      for (int d = 0; d < 3; ++d) Hll[d * 3 + d] += lambda;
      inverse3(Hll, Hinv);
      S Hinv_bl[3];
      for (int a = 0; a < 3; ++a)
        Hinv_bl[a] = Hinv[a * 3] * Jltr[0] + Hinv[a * 3 + 1] * Jltr[1] + Hinv[a * 3 + 2] * Jltr[2];
      // W_i = Jp_i^T Jl_i  (9x3)
      std::vector<S> W(size_t(K) * 27, S(0));
      for (int i = 0; i < K; ++i)
        for (int a = 0; a < P; ++a)
          for (int c = 0; c < 3; ++c)
            W[i * 27 + a * 3 + c] = Jp[(2 * i) * P + a] * Jl[(2 * i) * 3 + c] +
                                    Jp[(2 * i + 1) * P + a] * Jl[(2 * i + 1) * 3 + c];
      for (int i = 0; i < K; ++i) {
        const int ci = obs_cam_[lm_off_[l] + i];
        if (H) {
          for (int a = 0; a < P; ++a)
            for (int bb = 0; bb < P; ++bb)
              (*H)[(size_t(P) * ci + a) * n + P * ci + bb] +=
                  Jp[(2 * i) * P + a] * Jp[(2 * i) * P + bb] +
                  Jp[(2 * i + 1) * P + a] * Jp[(2 * i + 1) * P + bb];
          for (int j = 0; j < K; ++j) {
            const int cj = obs_cam_[lm_off_[l] + j];
            for (int a = 0; a < P; ++a) {
              S wa[3];
              for (int c = 0; c < 3; ++c)
                wa[c] = W[i * 27 + a * 3 + 0] * Hinv[0 * 3 + c] +
                        W[i * 27 + a * 3 + 1] * Hinv[1 * 3 + c] +
                        W[i * 27 + a * 3 + 2] * Hinv[2 * 3 + c];
              for (int bb = 0; bb < P; ++bb)
                (*H)[(size_t(P) * ci + a) * n + P * cj + bb] -=
                    wa[0] * W[j * 27 + bb * 3 + 0] + wa[1] * W[j * 27 + bb * 3 + 1] +
                    wa[2] * W[j * 27 + bb * 3 + 2];
            }
          }
        }
        for (int a = 0; a < P; ++a) {
          S v = 0;
          for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * i + rr;
            const S t = r[row] - (Jl[row * 3] * Hinv_bl[0] + Jl[row * 3 + 1] * Hinv_bl[1] +
                                  Jl[row * 3 + 2] * Hinv_bl[2]);
            v += Jp[row * P + a] * t;
          }
          b[P * ci + a] += v;
        }
      }
    }
    if (H && pose_lambda > 0)
      for (size_t i = 0; i < n; ++i) (*H)[i * n + i] += pose_lambda;
  }

  // LandmarkBlockSC::back_substitute (sc/landmark_block.hpp:409-446) on the
  // current state; updates landmarks, returns l_diff.
  S sc_back_substitute(S lambda, const S* pose_scaling, const S* pose_inc) {
    double l_diff = 0;
    for (int l = 0; l < n_lms_; ++l) {
      const int K = k(l);
      std::vector<S> Jp(size_t(2 * K) * P), Jl(size_t(2 * K) * 3), r(2 * K);
      S jls[3];
      sc_linearize(l, pose_scaling, Jp, Jl, r, nullptr, jls);
      S Hll[9] = {0}, tmp[3] = {0};
      std::vector<S> J_inc(2 * K, S(0));
      for (int i = 0; i < K; ++i) {
        const int ci = obs_cam_[lm_off_[l] + i];
        for (int rr = 0; rr < 2; ++rr) {
          const int row = 2 * i + rr;
          S jp_inc = 0;
          for (int a = 0; a < P; ++a) jp_inc += Jp[row * P + a] * pose_inc[P * ci + a];
          J_inc[row] += jp_inc;
          for (int a = 0; a < 3; ++a) {
            tmp[a] += Jl[row * 3 + a] * (r[row] + jp_inc);
            for (int c = 0; c < 3; ++c) Hll[a * 3 + c] += Jl[row * 3 + a] * Jl[row * 3 + c];
          }
        }
      }
      for (int d = 0; d < 3; ++d) Hll[d * 3 + d] += lambda;
      S Hinv[9];
      inverse3(Hll, Hinv);
      S inc[3];
      for (int a = 0; a < 3; ++a)
        inc[a] = -(Hinv[a * 3] * tmp[0] + Hinv[a * 3 + 1] * tmp[1] + Hinv[a * 3 + 2] * tmp[2]);
      S acc = 0;
      for (int row = 0; row < 2 * K; ++row) {
        J_inc[row] += Jl[row * 3] * inc[0] + Jl[row * 3 + 1] * inc[1] + Jl[row * 3 + 2] * inc[2];
        acc += J_inc[row] * (S(0.5) * J_inc[row] + r[row]);
      }
      l_diff -= double(acc);
      for (int c = 0; c < 3; ++c) lms_[3 * l + c] += inc[c] * jls[c];
    }
    return S(l_diff);
  }

 private:
  // LandmarkBlockSC::linearize_landmark + scale_Jl_cols + scale_Jp_cols
  // (sc/landmark_block.hpp:127-213)
  void sc_linearize(int l, const S* pose_scaling, std::vector<S>& Jp,
                    std::vector<S>& Jl, std::vector<S>& r, S* jp_diag2,
                    S* jl_scale_out) const {
    const int K = k(l);
    std::fill(Jp.begin(), Jp.end(), S(0));
    std::fill(Jl.begin(), Jl.end(), S(0));
    std::fill(r.begin(), r.end(), S(0));
    for (int i = 0; i < K; ++i) {
      const int64_t o = lm_off_[l] + i;
      S jp[12], ji[6], jl[6], res[2];
      const bool valid = linearize_point<S>(&obs_xy_[2 * o], &lms_[3 * l],
                                            &cams_[10 * obs_cam_[o]], true, res, jp, ji, jl);
      if (!opt_.use_valid_projections_only || valid) {
        const S res_squared = res[0] * res[0] + res[1] * res[1];
        S werr, w;
        compute_error_weight<S>(opt_, res_squared, werr, w);
        const S sw = std::sqrt(w);
        for (int rr = 0; rr < 2; ++rr) {
          const int row = 2 * i + rr;
          for (int c = 0; c < 6; ++c) Jp[row * P + c] = sw * jp[6 * rr + c];
          for (int c = 0; c < 3; ++c) Jp[row * P + 6 + c] = sw * ji[3 * rr + c];
          for (int c = 0; c < 3; ++c) Jl[row * 3 + c] = sw * jl[3 * rr + c];
          r[row] = sw * res[rr];
        }
      }
    }
    if (jp_diag2) {
      for (int i = 0; i < K; ++i) {
        const int ci = obs_cam_[lm_off_[l] + i];
        for (int a = 0; a < P; ++a)
          jp_diag2[P * ci + a] += Jp[(2 * i) * P + a] * Jp[(2 * i) * P + a] +
                                  Jp[(2 * i + 1) * P + a] * Jp[(2 * i + 1) * P + a];
      }
    }
    for (int c = 0; c < 3; ++c) {
      S s2 = 0;
      for (int row = 0; row < 2 * K; ++row) s2 += Jl[row * 3 + c] * Jl[row * 3 + c];
      const S sc = S(1) / (eps_ + std::sqrt(s2));
      if (jl_scale_out) jl_scale_out[c] = sc;
      for (int row = 0; row < 2 * K; ++row) Jl[row * 3 + c] *= sc;
    }
    if (pose_scaling) {
      for (int i = 0; i < K; ++i) {
        const int ci = obs_cam_[lm_off_[l] + i];
        for (int rr = 0; rr < 2; ++rr)
          for (int a = 0; a < P; ++a) Jp[(2 * i + rr) * P + a] *= pose_scaling[P * ci + a];
      }
    }
  }

  int n_cams_, n_lms_;
  int64_t n_obs_ = 0;
  int n_threads_ = 1;
  Options opt_;
  S eps_;
  std::vector<int64_t> lm_off_;
  std::vector<int32_t> obs_cam_;
  std::vector<S> obs_xy_;
  std::vector<S> cams_, lms_, cams_bak_, lms_bak_;
  std::vector<size_t> blk_off_;
  struct FreeDeleter {
    void operator()(S* p) const { std::free(p); }
  };
  std::unique_ptr<S[], FreeDeleter> storage_;
  size_t storage_size_ = 0;
  mutable std::vector<std::vector<S>> acc_;  // [thread][9 n_c]
  mutable std::vector<std::vector<S>> xr_, tmp_;  // [thread] gather / row-product scratch (no per-call allocation)
  std::vector<S>& xr_scratch(int tid) const { return xr_[tid]; }
  std::vector<S>& tmp_scratch(int tid) const { return tmp_[tid]; }
  std::vector<S> jl_col_scale_;
  std::vector<S> rot_;
  std::vector<char> damped_, failed_;
  S pose_damping_ = 0;
  std::vector<S> jp_diag2_, pose_scaling_, precond_blocks_, inv_blocks_, b_;
  std::vector<S> pw_Jp_, pw_Jl_, pw_Hll_inv_;
  mutable std::vector<S> sc_H_;  // dense reduced camera matrix of the SC solver mode
  std::vector<S> mf_Hpp_;        // solver_type 2: the diagonal blocks Hpp + lambda I
  S sc_lambda_ = 0;
  bool new_linearization_point_ = false;
};

}  // namespace orc
