// -*- c++ -*-
// Stand-in for Sophus::SE3 (see sophus/common.hpp): (SO3, translation); parameters are the 7 numbers
// (qx, qy, qz, qw, tx, ty, tz) as in Sophus; T1 * T2 = (R1 R2, t1 + R1 t2); inverse = (R^-1, -(R^-1 t)).
#pragma once
#include "so3.hpp"

namespace Sophus {
template <class S>
class SE3 {
 public:
  using Scalar = S;
  using Vec3 = Eigen::Matrix<S, 3, 1>;
  using Vec7 = Eigen::Matrix<S, 7, 1>;
  using Mat3 = Eigen::Matrix<S, 3, 3>;
  using Mat4 = Eigen::Matrix<S, 4, 4>;
  using Tangent = Eigen::Matrix<S, 6, 1>;
  static constexpr int num_parameters = 7;
  static constexpr int DoF = 6;
  SE3() { t_.setZero(); }
  SE3(const SO3<S>& so3, const Vec3& t) : so3_(so3), t_(t) {}
  template <class D>
  SE3(const SO3<S>& so3, const Eigen::MatrixBase<D>& t) : so3_(so3), t_(t) {}
  SO3<S>& so3() { return so3_; }
  const SO3<S>& so3() const { return so3_; }
  Vec3& translation() { return t_; }
  const Vec3& translation() const { return t_; }
  Mat3 rotationMatrix() const { return so3_.matrix(); }
  Mat4 matrix() const {
    Mat4 m;
    m.setIdentity();
    m.template topLeftCorner<3, 3>() = so3_.matrix();
    m.template topRightCorner<3, 1>() = t_;
    return m;
  }
  Eigen::Matrix<S, 3, 4> matrix3x4() const {
    Eigen::Matrix<S, 3, 4> m;
    m.template topLeftCorner<3, 3>() = so3_.matrix();
    m.template topRightCorner<3, 1>() = t_;
    return m;
  }
  SE3 inverse() const {
    const SO3<S> inv = so3_.inverse();
    return SE3(inv, inv * (t_ * S(-1)));
  }
  SE3 operator*(const SE3& o) const { return SE3(so3_ * o.so3_, t_ + so3_ * o.t_); }
  SE3& operator*=(const SE3& o) {
    *this = *this * o;
    return *this;
  }
  Vec3 operator*(const Vec3& p) const { return so3_ * p + t_; }
  Vec7 params() const {
    Vec7 p;
    p.template head<4>() = so3_.params();
    p.template tail<3>() = t_;
    return p;
  }
  template <class T>
  SE3<T> cast() const {
    return SE3<T>(so3_.template cast<T>(), t_.template cast<T>());
  }

 private:
  SO3<S> so3_;
  Vec3 t_;
};
using SE3d = SE3<double>;
using SE3f = SE3<float>;
}  // namespace Sophus

namespace Eigen {
// Eigen::Map<Sophus::SE3<S> const>(ptr): the 7 parameters are taken as they are (no normalisation)
template <class S, int MO, class St>
class Map<const Sophus::SE3<S>, MO, St> {
 public:
  explicit Map(const S* p) : p_(p) {}
  operator Sophus::SE3<S>() const {
    return Sophus::SE3<S>(Sophus::SO3<S>::from_raw(p_[0], p_[1], p_[2], p_[3]),
                          Eigen::Matrix<S, 3, 1>(p_[4], p_[5], p_[6]));
  }

 private:
  const S* p_;
};
}  // namespace Eigen
