// -*- c++ -*-
// Stand-in for the part of Sophus (un-vendored submodule of the reference) that the
// reference's hot path touches. TEST INFRASTRUCTURE ONLY - see Eigen/Dense in this directory.
// Restated from Sophus' published definitions:
//   Constants<Scalar>::epsilon()  = 1e-10 (double) / 1e-5 (float);  epsilonSqrt() = sqrt(epsilon())
#pragma once
#include <cmath>

#include <Eigen/Dense>

namespace Sophus {
template <class Scalar>
struct Constants {
  static Scalar epsilon() { return Scalar(1e-10); }
  static Scalar epsilonSqrt() {
    using std::sqrt;
    return sqrt(epsilon());
  }
  static Scalar pi() { return Scalar(3.141592653589793238462643383279502884); }
};
template <>
struct Constants<float> {
  static float constexpr epsilon() { return static_cast<float>(1e-5); }
  static float epsilonSqrt() { return std::sqrt(epsilon()); }
  static float constexpr pi() { return 3.141592653589793238462643383279502884f; }
};
template <class S, int N>
using Vector = Eigen::Matrix<S, N, 1>;
template <class S>
using Vector3 = Eigen::Matrix<S, 3, 1>;
template <class S>
using Vector6 = Eigen::Matrix<S, 6, 1>;
template <class S>
using Matrix3 = Eigen::Matrix<S, 3, 3>;
template <class S>
using Matrix4 = Eigen::Matrix<S, 4, 4>;
}  // namespace Sophus
