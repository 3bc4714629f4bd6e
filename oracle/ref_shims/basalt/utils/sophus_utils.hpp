// -*- c++ -*-
// Stand-in for the one helper of basalt-headers' sophus_utils.hpp that the reference uses
// (TEST INFRASTRUCTURE ONLY). se3_expd is basalt's DECOUPLED exponential: rotation = SO3::exp(omega),
// translation = upsilon taken as it is (reference call site: src/rootba/bal/bal_problem.hpp:99-101).
#pragma once
#include <sophus/se3.hpp>
namespace Sophus {
template <class Derived>
inline SE3<typename Derived::Scalar> se3_expd(const Eigen::MatrixBase<Derived>& upsilon_omega) {
  using Scalar = typename Derived::Scalar;
  Eigen::Matrix<Scalar, 3, 1> ups = upsilon_omega.template head<3>();
  Eigen::Matrix<Scalar, 3, 1> om = upsilon_omega.template tail<3>();
  return SE3<Scalar>(SO3<Scalar>::exp(om), ups);
}
}  // namespace Sophus
