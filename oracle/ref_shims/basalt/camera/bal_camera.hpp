// -*- c++ -*-
// Stand-in for basalt::BalCamera (basalt-headers, un-vendored submodule of the reference, SHA unknown;
// SURVEY.md 8c). TEST INFRASTRUCTURE ONLY - see Eigen/Dense in this directory.
// Restated from the published model ("Bundle Adjustment in the Large" camera with the rootba sign
// convention: z forward, no minus): m = p.xy / p.z, r2 = |m|^2, rp = 1 + k1 r2 + k2 r2^2,
// proj = f rp m, valid iff z >= Sophus::Constants<Scalar>::epsilonSqrt(); analytic Jacobians w.r.t.
// the point (2x4, last column zero) and w.r.t. (f, k1, k2). The reference pins the projection value
// against its in-tree formula (src/rootba/bal/snavely_projection.test.cpp:155-188) and the Jacobians by
// numeric differentiation (bal_bundle_adjustment_helper.test.cpp:54-148); oracle/ref_driver.cpp repeats
// the latter check on this stand-in.
#pragma once
#include <cstddef>
#include <type_traits>

#include <basalt/utils/assert.h>
#include <basalt/utils/sophus_utils.hpp>

namespace basalt {
template <class Scalar_>
class BalCamera {
 public:
  using Scalar = Scalar_;
  static constexpr int N = 3;
  using Vec2 = Eigen::Matrix<Scalar, 2, 1>;
  using Vec4 = Eigen::Matrix<Scalar, 4, 1>;
  using VecN = Eigen::Matrix<Scalar, N, 1>;
  using Mat24 = Eigen::Matrix<Scalar, 2, 4>;
  using Mat2N = Eigen::Matrix<Scalar, 2, N>;

  BalCamera() { param_.setZero(); }
  explicit BalCamera(const VecN& p) : param_(p) {}
  template <class D>
  explicit BalCamera(const Eigen::MatrixBase<D>& p) : param_(p) {}
  template <class Scalar2>
  BalCamera<Scalar2> cast() const {
    return BalCamera<Scalar2>(param_.template cast<Scalar2>());
  }
  static std::string getName() { return "bal"; }
  const VecN& getParam() const { return param_; }
  void operator+=(const VecN& inc) { param_ += inc; }
  template <class D>
  void operator+=(const Eigen::MatrixBase<D>& inc) {
    param_ += inc;
  }

  template <class DerivedPoint3D, class DerivedPoint2D, class DerivedJ3D = std::nullptr_t,
            class DerivedJparam = std::nullptr_t>
  inline bool project(const Eigen::MatrixBase<DerivedPoint3D>& p3d, Eigen::MatrixBase<DerivedPoint2D>& proj,
                      DerivedJ3D d_proj_d_p3d = nullptr, DerivedJparam d_proj_d_param = nullptr) const {
    const Scalar f = param_[0], k1 = param_[1], k2 = param_[2];
    const Scalar x = p3d[0], y = p3d[1], z = p3d[2];
    const Scalar mx = x / z;
    const Scalar my = y / z;
    const Scalar mx2 = mx * mx;
    const Scalar my2 = my * my;
    const Scalar r2 = mx2 + my2;
    const Scalar r4 = r2 * r2;
    const Scalar rp = Scalar(1) + k1 * r2 + k2 * r4;
    proj.derived() = Vec2(f * mx * rp, f * my * rp);
    const bool is_valid = z >= Sophus::Constants<Scalar>::epsilonSqrt();
    if constexpr (!std::is_same_v<DerivedJ3D, std::nullptr_t>) {
      if (d_proj_d_p3d) {
        d_proj_d_p3d->setZero();
        const Scalar tmp = k1 + k2 * Scalar(2) * r2;
        (*d_proj_d_p3d)(0, 0) = f * (rp + Scalar(2) * mx2 * tmp) / z;
        (*d_proj_d_p3d)(1, 1) = f * (rp + Scalar(2) * my2 * tmp) / z;
        (*d_proj_d_p3d)(1, 0) = (*d_proj_d_p3d)(0, 1) = f * my * mx * Scalar(2) * tmp / z;
        (*d_proj_d_p3d)(0, 2) = -f * mx * (rp + Scalar(2) * tmp * r2) / z;
        (*d_proj_d_p3d)(1, 2) = -f * my * (rp + Scalar(2) * tmp * r2) / z;
      }
    }
    if constexpr (!std::is_same_v<DerivedJparam, std::nullptr_t>) {
      if (d_proj_d_param) {
        d_proj_d_param->setZero();
        (*d_proj_d_param)(0, 0) = mx * rp;
        (*d_proj_d_param)(0, 1) = f * mx * r2;
        (*d_proj_d_param)(0, 2) = f * mx * r4;
        (*d_proj_d_param)(1, 0) = my * rp;
        (*d_proj_d_param)(1, 1) = f * my * r2;
        (*d_proj_d_param)(1, 2) = f * my * r4;
      }
    }
    return is_valid;
  }

 private:
  VecN param_;
};
}  // namespace basalt
