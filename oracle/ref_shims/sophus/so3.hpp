// -*- c++ -*-
// Stand-in for Sophus::SO3 (see sophus/common.hpp). Restated from Sophus' published algorithms:
//   storage: unit quaternion, coefficient order (x, y, z, w)
//   exp(omega): q = (cos(theta/2), sin(theta/2)/theta * omega), Taylor branch for theta^2 < eps^2
//   constructor from a quaternion normalises (q /= |q|); the group product goes through it
//   R * p: p + w t + q_v x t with t = 2 q_v x p (Eigen's quaternion rotation)
//   matrix(): Eigen's Quaternion::toRotationMatrix
//   SO3(R): Eigen's rotation-matrix-to-quaternion (largest-diagonal branch), then normalise
#pragma once
#include "common.hpp"

namespace Sophus {
template <class S>
class SO3 {
 public:
  using Scalar = S;
  using Vec3 = Eigen::Matrix<S, 3, 1>;
  using Vec4 = Eigen::Matrix<S, 4, 1>;
  using Mat3 = Eigen::Matrix<S, 3, 3>;
  using Tangent = Vec3;
  using Point = Vec3;
  using Transformation = Mat3;
  static constexpr int num_parameters = 4;
  static constexpr int DoF = 3;

  SO3() { q_ = Vec4(S(0), S(0), S(0), S(1)); }
  SO3(const SO3&) = default;
  SO3& operator=(const SO3&) = default;
  // from quaternion coefficients (x, y, z, w); normalises like Sophus' quaternion constructor
  static SO3 from_quaternion(S x, S y, S z, S w) {
    SO3 r;
    r.q_ = Vec4(x, y, z, w);
    r.normalize();
    return r;
  }
  // raw coefficients without normalisation (what Eigen::Map<SO3 const> hands over)
  static SO3 from_raw(S x, S y, S z, S w) {
    SO3 r;
    r.q_ = Vec4(x, y, z, w);
    return r;
  }
  template <class D>
  explicit SO3(const Eigen::MatrixBase<D>& Rm) {
    Mat3 R(Rm);
    // Eigen/src/Geometry/Quaternion.h quaternionbase_assign_impl<Other,3,3>
    using std::sqrt;
    S t = R(0, 0) + R(1, 1) + R(2, 2);
    S x, y, z, w;
    if (t > S(0)) {
      t = sqrt(t + S(1));
      w = S(0.5) * t;
      t = S(0.5) / t;
      x = (R(2, 1) - R(1, 2)) * t;
      y = (R(0, 2) - R(2, 0)) * t;
      z = (R(1, 0) - R(0, 1)) * t;
    } else {
      int i = 0;
      if (R(1, 1) > R(0, 0)) i = 1;
      if (R(2, 2) > R(i, i)) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      t = sqrt(R(i, i) - R(j, j) - R(k, k) + S(1));
      S v[3];
      v[i] = S(0.5) * t;
      t = S(0.5) / t;
      w = (R(k, j) - R(j, k)) * t;
      v[j] = (R(j, i) + R(i, j)) * t;
      v[k] = (R(k, i) + R(i, k)) * t;
      x = v[0];
      y = v[1];
      z = v[2];
    }
    q_ = Vec4(x, y, z, w);
    normalize();
  }
  void normalize() {
    const S n = q_.norm();
    q_ /= n;
  }
  static SO3 exp(const Vec3& omega) {
    using std::cos;
    using std::sin;
    using std::sqrt;
    const S theta_sq = omega.squaredNorm();
    S imag_factor, real_factor;
    if (theta_sq < Constants<S>::epsilon() * Constants<S>::epsilon()) {
      const S theta_po4 = theta_sq * theta_sq;
      imag_factor = S(0.5) - S(1.0 / 48.0) * theta_sq + S(1.0 / 3840.0) * theta_po4;
      real_factor = S(1) - S(1.0 / 8.0) * theta_sq + S(1.0 / 384.0) * theta_po4;
    } else {
      const S theta = sqrt(theta_sq);
      const S half_theta = S(0.5) * theta;
      const S sin_half_theta = sin(half_theta);
      imag_factor = sin_half_theta / theta;
      real_factor = cos(half_theta);
    }
    // (Sophus builds the SO3 from this quaternion without a further normalisation beyond its
    //  unit-length check)
    return from_raw(imag_factor * omega(0), imag_factor * omega(1), imag_factor * omega(2), real_factor);
  }
  template <class D>
  static SO3 exp(const Eigen::MatrixBase<D>& omega) {
    return exp(Vec3(omega));
  }
  static Mat3 hat(const Vec3& o) {
    Mat3 m;
    m(0, 0) = S(0);
    m(0, 1) = -o(2);
    m(0, 2) = o(1);
    m(1, 0) = o(2);
    m(1, 1) = S(0);
    m(1, 2) = -o(0);
    m(2, 0) = -o(1);
    m(2, 1) = o(0);
    m(2, 2) = S(0);
    return m;
  }
  template <class D>
  static Mat3 hat(const Eigen::MatrixBase<D>& o) {
    return hat(Vec3(o));
  }
  SO3 inverse() const { return from_raw(-q_(0), -q_(1), -q_(2), q_(3)); }
  SO3 operator*(const SO3& o) const {
    const S ax = q_(0), ay = q_(1), az = q_(2), aw = q_(3);
    const S bx = o.q_(0), by = o.q_(1), bz = o.q_(2), bw = o.q_(3);
    return from_quaternion(aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                           aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz);
  }
  SO3& operator*=(const SO3& o) {
    *this = *this * o;
    return *this;
  }
  Vec3 operator*(const Vec3& p) const {
    const Vec3 qv(q_(0), q_(1), q_(2));
    Vec3 uv = qv.cross(p);
    uv += uv;
    return p + q_(3) * uv + qv.cross(uv);
  }
  template <class D>
  Vec3 operator*(const Eigen::MatrixBase<D>& p) const {
    return *this * Vec3(p);
  }
  Mat3 matrix() const {
    const S x = q_(0), y = q_(1), z = q_(2), w = q_(3);
    const S tx = S(2) * x, ty = S(2) * y, tz = S(2) * z;
    const S twx = tx * w, twy = ty * w, twz = tz * w;
    const S txx = tx * x, txy = ty * x, txz = tz * x;
    const S tyy = ty * y, tyz = tz * y, tzz = tz * z;
    Mat3 R;
    R(0, 0) = S(1) - (tyy + tzz);
    R(0, 1) = txy - twz;
    R(0, 2) = txz + twy;
    R(1, 0) = txy + twz;
    R(1, 1) = S(1) - (txx + tzz);
    R(1, 2) = tyz - twx;
    R(2, 0) = txz - twy;
    R(2, 1) = tyz + twx;
    R(2, 2) = S(1) - (txx + tyy);
    return R;
  }
  const Vec4& params() const { return q_; }
  const Vec4& coeffs_xyzw() const { return q_; }
  template <class T>
  SO3<T> cast() const {
    return SO3<T>::from_raw(T(q_(0)), T(q_(1)), T(q_(2)), T(q_(3)));
  }

 private:
  Vec4 q_;
};
using SO3d = SO3<double>;
using SO3f = SO3<float>;
}  // namespace Sophus
