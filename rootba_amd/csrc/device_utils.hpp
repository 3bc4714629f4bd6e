// device_utils.hpp — wave64 primitives and per-observation geometry for gfx950.
//
// MI355X-native building blocks shared by all landmark kernels:
//  * wave_sum(): 64-lane reduction on the DPP network (no LDS, no shuffles
//    through the LDS crossbar), result broadcast through an SGPR readlane;
//  * hardware floating-point atomics for the camera-indexed scatter-adds;
//  * the BAL/Snavely projection with analytic Jacobians, i.e. what the reference
//    evaluates per observation in BalBundleAdjustmentHelper::linearize_point
//    (reference src/rootba/bal/bal_bundle_adjustment_helper.cpp:111-149, camera
//    model basalt::BalCamera) — restated for one GPU lane.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

namespace rba {

constexpr int P = 9;  // POSE_SIZE: 6 pose + 3 intrinsics (linearizor_qr.hpp:51)

// ---------------------------------------------------------------------------
// DPP wave reduction (wave64). Steps: quad_perm[1,0,3,2], quad_perm[2,3,0,1],
// row_ror:4, row_ror:8 give every lane its 16-lane row sum; row_bcast:15 and
// row_bcast:31 fold the four rows into lane 63.
// MUST be called with all 64 lanes active (wave-uniform control flow).
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov0(float v) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov0(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, int(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, int(b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double((static_cast<long long>(hi) << 32) |
                              static_cast<unsigned int>(lo));
}

__device__ __forceinline__ float read_lane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double read_lane(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane(int(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane(int(b >> 32), lane);
  return __longlong_as_double((static_cast<long long>(hi) << 32) |
                              static_cast<unsigned int>(lo));
}

// sum over the 16 lanes of a DPP row (the first four steps of wave_sum); every lane of
// the row receives it
template <class S>
__device__ __forceinline__ S row_sum(S v) {
  v += dpp_mov0<0xb1>(v);   // quad_perm:[1,0,3,2]
  v += dpp_mov0<0x4e>(v);   // quad_perm:[2,3,0,1]
  v += dpp_mov0<0x124>(v);  // row_ror:4
  v += dpp_mov0<0x128>(v);  // row_ror:8
  return v;
}

template <class S>
__device__ __forceinline__ S wave_sum(S v) {
  v += dpp_mov0<0xb1>(v);   // quad_perm:[1,0,3,2]
  v += dpp_mov0<0x4e>(v);   // quad_perm:[2,3,0,1]
  v += dpp_mov0<0x124>(v);  // row_ror:4
  v += dpp_mov0<0x128>(v);  // row_ror:8
  v += dpp_mov0<0x142>(v);  // row_bcast:15
  v += dpp_mov0<0x143>(v);  // row_bcast:31
  return read_lane(v, 63);
}

// compiler-level ordering of LDS traffic inside ONE wave (the LDS itself
// executes a wave's DS instructions in issue order)
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// hardware FP atomics (global_atomic_add_f32 / _f64), no CAS loop
__device__ __forceinline__ void atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }

// the same on an LDS address (ds_add_f32 / ds_add_f64, no return value); the pointer must be
// derived from a __shared__ object so that the compiler sees the LDS address space
__device__ __forceinline__ void lds_atomic_add(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_atomic_add(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <class S>
struct Eps;
// Sophus::Constants<Scalar>::epsilon() / epsilonSqrt()
template <>
struct Eps<float> {
  static constexpr float eps = 1e-5f;
  static constexpr float eps_sqrt = 3.1622776601683794e-3f;
  static constexpr float tiny = 1.17549435e-38f;  // FLT_MIN (Eigen makeHouseholder tol)
};
template <>
struct Eps<double> {
  static constexpr double eps = 1e-10;
  static constexpr double eps_sqrt = 1e-5;
  static constexpr double tiny = 2.2250738585072014e-308;
};

template <class S>
__device__ __forceinline__ bool is_finite(S v) {
  return isfinite(v);
}

// 1-ulp reciprocal / square root of the hardware (v_rcp_f32, v_sqrt_f32) for float; exact operations for double. The
// IEEE-correct float division and square root the compiler emits by default are ten-instruction sequences (scale,
// reciprocal, four FMAs, fix-up). ONLY for pure scalings that are stored and re-used consistently (the Jl column scale
// of the QR pass): used for the reflector scalars (beta, 1 / (c0 - beta), tau) they cost orthogonality of
// H = I - tau v v^T and, on an ill-conditioned final-13682 state, accuracy of a long matrix-free solve (kernels_s1.hpp).
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double fast_rcp(double x) { return 1.0 / x; }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double fast_sqrt(double x) { return sqrt(x); }

// unit quaternion (x,y,z,w) -> rotation matrix, row-major
template <class S>
__device__ __forceinline__ void quat_to_rot(S x, S y, S z, S w, S R[9]) {
  const S tx = S(2) * x, ty = S(2) * y, tz = S(2) * z;
  const S twx = tx * w, twy = ty * w, twz = tz * w;
  const S txx = tx * x, txy = ty * x, txz = tz * x;
  const S tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = S(1) - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = S(1) - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = S(1) - (txx + tyy);
}

// The camera-frame point p_c = R(q) p_w + t and R itself.
// float: evaluated in DOUBLE from the float state and rounded once (round 6). R p_w and t are both of the size of the
// scene and cancel down to the depth: in float arithmetic p_c carries an ABSOLUTE error of ~1e-7 |p_w| whatever its own
// size, and a landmark that an LM step has moved next to a camera plane (final-13682, state of LM iteration 6: two
// 2-observation landmarks at depth 0.008 with |p_w| = 126, residuals of thousands of pixels) then has residual and
// Jacobian rows that are wrong by 1e-3 - two such landmarks were 3e-5 of the whole cost and a third of the increment
// error of that state, for the float32 CPU restatement of the reference (1.1e-3 from float64, 28 % of it in ONE camera)
// as for this library (8.4e-3; profiles/r6_final13682_iteration6_diagnosis.txt). ~40 double-precision operations per
// observation in kernels that wait for HBM; everything after p_c - the projection, its Jacobians, the QR - is float
// arithmetic on well-scaled quantities as in the reference (bal_bundle_adjustment_helper.cpp:111-149).
template <class S>
__device__ __forceinline__ void camera_frame_point(const S* __restrict__ cam, S pwx, S pwy, S pwz, S R[9], S& px, S& py,
                                                   S& pz) {
  double Rd[9];
  quat_to_rot<double>(double(cam[0]), double(cam[1]), double(cam[2]), double(cam[3]), Rd);
  const double x = double(pwx), y = double(pwy), z = double(pwz);
  px = S(Rd[0] * x + Rd[1] * y + Rd[2] * z + double(cam[4]));
  py = S(Rd[3] * x + Rd[4] * y + Rd[5] * z + double(cam[5]));
  pz = S(Rd[6] * x + Rd[7] * y + Rd[8] * z + double(cam[6]));
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = S(Rd[i]);
}

// Residual only (compute_error path). cam = (q xyzw, t, f, k1, k2).
template <class S>
__device__ __forceinline__ bool project_residual(const S* __restrict__ cam, S pwx, S pwy, S pwz,
                                                 S ox, S oy, S& rx, S& ry) {
  S R[9], px, py, pz;
  camera_frame_point<S>(cam, pwx, pwy, pwz, R, px, py, pz);
  const S mx = px / pz, my = py / pz;
  const S r2 = mx * mx + my * my;
  const S rp = S(1) + cam[8] * r2 + cam[9] * r2 * r2;
  rx = cam[7] * mx * rp - ox;
  ry = cam[7] * my * rp - oy;
  return pz >= Eps<S>::eps_sqrt;
}

// Huber / trivial loss: weighted error and weight
// (compute_error_weight, bal_bundle_adjustment_helper.cpp:43-66)
template <class S>
__device__ __forceinline__ void error_weight(int robust_norm, S huber, S res_sq, S& err, S& w) {
  if (robust_norm == 1) {
    w = res_sq < huber * huber ? S(1) : huber / sqrt(res_sq);
    err = S(0.5) * (S(2) - w) * w * res_sq;
  } else {
    w = S(1);
    err = S(0.5) * res_sq;
  }
}

// Full linearisation of one observation.
// Jp: 2x9 row-major ([pose(6) | intrinsics(3)]), Jl: 2x3, res: 2.
template <class S>
__device__ __forceinline__ bool linearize_obs(const S* __restrict__ cam, S pwx, S pwy, S pwz,
                                              S ox, S oy, S res[2], S Jp[18], S Jl[6]) {
  S R[9], px, py, pz;
  camera_frame_point<S>(cam, pwx, pwy, pwz, R, px, py, pz);
  const S f = cam[7], k1 = cam[8], k2 = cam[9];
  const S iz = S(1) / pz;
  const S mx = px * iz, my = py * iz;
  const S r2 = mx * mx + my * my;
  const S r4 = r2 * r2;
  const S rp = S(1) + k1 * r2 + k2 * r4;
  res[0] = f * mx * rp - ox;
  res[1] = f * my * rp - oy;
  const S tmp = k1 + S(2) * k2 * r2;
  // d proj / d p_cam
  S J[6];
  J[0] = f * (rp + S(2) * mx * mx * tmp) * iz;
  J[1] = S(2) * f * mx * my * tmp * iz;
  J[2] = -f * mx * (rp + S(2) * r2 * tmp) * iz;
  J[3] = J[1];
  J[4] = f * (rp + S(2) * my * my * tmp) * iz;
  J[5] = -f * my * (rp + S(2) * r2 * tmp) * iz;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const S j0 = J[3 * r], j1 = J[3 * r + 1], j2 = J[3 * r + 2];
    // pose part: [I | -hat(p_cam)]
    Jp[9 * r + 0] = j0;
    Jp[9 * r + 1] = j1;
    Jp[9 * r + 2] = j2;
    Jp[9 * r + 3] = j2 * py - j1 * pz;
    Jp[9 * r + 4] = j0 * pz - j2 * px;
    Jp[9 * r + 5] = j1 * px - j0 * py;
    // landmark part: J * R
    Jl[3 * r + 0] = j0 * R[0] + j1 * R[3] + j2 * R[6];
    Jl[3 * r + 1] = j0 * R[1] + j1 * R[4] + j2 * R[7];
    Jl[3 * r + 2] = j0 * R[2] + j1 * R[5] + j2 * R[8];
  }
  // intrinsics part (f, k1, k2)
  Jp[6] = mx * rp;
  Jp[7] = f * mx * r2;
  Jp[8] = f * mx * r4;
  Jp[15] = my * rp;
  Jp[16] = f * my * r2;
  Jp[17] = f * my * r4;
  return pz >= Eps<S>::eps_sqrt;
}

// Eigen-style Givens coefficients: G^T [p; q] = [r; 0] (Eigen's makeGivens: the larger of |p|, |q|
// is the divisor). Branch-free: the two general cases differ only in which operand is the divisor and where the
// results go, so ONE division, one square root and one reciprocal serve both (as four-way branches the lanes of a wave
// - neighbouring landmarks - diverge and every wave pays for both cases: 25 of the 70 division / square-root
// instructions per rotation; k_s2_obs evaluates six rotations per work-item). Every lane performs exactly the
// operations of its own case in the same order: the results are those of the branches bit for bit (up to the sign of a
// zero); q == 0 (which includes p == q == 0, where the general formula would divide zero by zero) is selected last.
template <class S>
__device__ __forceinline__ void make_givens(S p, S q, S& c, S& s) {
  const bool p_is_divisor = fabs(p) > fabs(q);
  const S den = p_is_divisor ? p : q, num = p_is_divisor ? q : p;
  const S t = num / den;
  S u = sqrt(S(1) + t * t);
  if (den < S(0)) u = -u;
  const S inv = S(1) / u;
  // |p| > |q|: c = 1 / u, s = -t c.        else: s = -1 / u, c = -t s.
  const S first = p_is_divisor ? inv : -inv;
  const S second = -t * first;
  c = p_is_divisor ? first : second;
  s = p_is_divisor ? second : first;
  if (q == S(0)) {
    c = p < S(0) ? S(-1) : S(1);
    s = S(0);
  }
}

}  // namespace rba
