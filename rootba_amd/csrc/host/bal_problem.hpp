// bal_problem.hpp — host-side BAL data model of the MI355X-native solver.
//
// Mirrors the parts of the reference's `rootba::BalProblem<Scalar>`
// (reference src/rootba/bal/bal_problem.hpp:61-234, bal_problem.cpp) that sit
// on either side of the accelerated path: BAL text loading (:190-282),
// normalisation (:428-469), perturbation (:508-554), depth filtering (:471-506),
// the flat camera state of `Camera::params()` (bal_problem.hpp:84-95) and the
// load pipeline order of `load_normalized_bal_problem` (:794-832).
// Own implementation on plain arrays (no Eigen/Sophus); one-off host work,
// outside the accelerated hot path.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rootba_hip {

struct BalDatasetOptions {  // reference src/rootba/bal/bal_dataset_options.hpp:40-100
  std::string input;
  bool normalize = true;
  double normalization_scale = 100.0;
  double rotation_sigma = 0.0;
  double translation_sigma = 0.0;
  double point_sigma = 0.0;
  int random_seed = 38401;
  double init_depth_threshold = 0.0;
  bool quiet = false;
};

namespace detail {
using Mat3 = std::array<double, 9>;
using Vec3 = std::array<double, 3>;

inline Mat3 quat_to_rot(const double* q) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  return {1 - 2 * (y * y + z * z), 2 * (x * y - z * w),     2 * (x * z + y * w),
          2 * (x * y + z * w),     1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
          2 * (x * z - y * w),     2 * (y * z + x * w),     1 - 2 * (x * x + y * y)};
}
inline void rot_to_quat(const Mat3& m, double* q) {
  const double tr = m[0] + m[4] + m[8];
  double x, y, z, w;
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    w = 0.25 * s; x = (m[7] - m[5]) / s; y = (m[2] - m[6]) / s; z = (m[3] - m[1]) / s;
  } else if (m[0] > m[4] && m[0] > m[8]) {
    const double s = std::sqrt(1.0 + m[0] - m[4] - m[8]) * 2;
    w = (m[7] - m[5]) / s; x = 0.25 * s; y = (m[1] + m[3]) / s; z = (m[2] + m[6]) / s;
  } else if (m[4] > m[8]) {
    const double s = std::sqrt(1.0 + m[4] - m[0] - m[8]) * 2;
    w = (m[2] - m[6]) / s; x = (m[1] + m[3]) / s; y = 0.25 * s; z = (m[5] + m[7]) / s;
  } else {
    const double s = std::sqrt(1.0 + m[8] - m[0] - m[4]) * 2;
    w = (m[3] - m[1]) / s; x = (m[2] + m[6]) / s; y = (m[5] + m[7]) / s; z = 0.25 * s;
  }
  const double n = std::sqrt(x * x + y * y + z * z + w * w) * (w < 0 ? -1.0 : 1.0);
  q[0] = x / n; q[1] = y / n; q[2] = z / n; q[3] = w / n;
}
inline Mat3 so3_exp(const Vec3& w) {  // Rodrigues
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double th = std::sqrt(th2);
  const double a = th < 1e-8 ? 1.0 - th2 / 6 : std::sin(th) / th;
  const double b = th < 1e-8 ? 0.5 - th2 / 24 : (1 - std::cos(th)) / th2;
  const Mat3 K = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  Mat3 R{};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double kk = 0;
      for (int l = 0; l < 3; ++l) kk += K[3 * i + l] * K[3 * l + j];
      R[3 * i + j] = (i == j ? 1.0 : 0.0) + a * K[3 * i + j] + b * kk;
    }
  return R;
}
inline Mat3 mul(const Mat3& A, const Mat3& B) {
  Mat3 C{};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      for (int l = 0; l < 3; ++l) C[3 * i + j] += A[3 * i + l] * B[3 * l + j];
  return C;
}
inline Vec3 mul(const Mat3& A, const Vec3& v) {
  return {A[0] * v[0] + A[1] * v[1] + A[2] * v[2], A[3] * v[0] + A[4] * v[1] + A[5] * v[2],
          A[6] * v[0] + A[7] * v[1] + A[8] * v[2]};
}
inline Vec3 mul_t(const Mat3& A, const Vec3& v) {  // A^T v
  return {A[0] * v[0] + A[3] * v[1] + A[6] * v[2], A[1] * v[0] + A[4] * v[1] + A[7] * v[2],
          A[2] * v[0] + A[5] * v[1] + A[8] * v[2]};
}
// element n/2 of the sorted data (the reference's `median_destructive`)
inline double median_upper(std::vector<double>& v) {
  auto mid = v.begin() + v.size() / 2;
  std::nth_element(v.begin(), mid, v.end());
  return *mid;
}
}  // namespace detail

template <class Scalar>
class BalProblem {
 public:
  static constexpr int CAM_STATE_SIZE = 10;  // qx qy qz qw tx ty tz f k1 k2

  struct Observation {
    int cam;
    Scalar x, y;
  };
  struct Landmark {
    std::array<Scalar, 3> p_w;
    std::vector<Observation> obs;  // ascending camera index (std::map order)
  };

  std::vector<std::array<Scalar, 10>> cameras;
  std::vector<Landmark> landmarks;

  int num_cameras() const { return int(cameras.size()); }
  int num_landmarks() const { return int(landmarks.size()); }
  int64_t num_observations() const {
    int64_t n = 0;
    for (const auto& l : landmarks) n += int64_t(l.obs.size());
    return n;
  }

  // BAL text format: header, observations (cam lm x y), 9 parameters per camera
  // (Rodrigues, t, f, k1, k2), 3 per landmark. The camera looks down -z with y up
  // in BAL; here +z forward / y down, so y and z axes are flipped on load.
  void load_bal(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f) throw std::runtime_error("Could not open '" + path + "'");
    auto fail = [&]() {
      std::fclose(f);
      throw std::runtime_error("Failed to parse '" + path + "'");
    };
    int nc, nl, no;
    if (std::fscanf(f, "%d %d %d", &nc, &nl, &no) != 3 || nc <= 0 || nl <= 0 || no <= 0) fail();
    cameras.assign(nc, {});
    landmarks.assign(nl, {});
    for (int i = 0; i < no; ++i) {
      int c, l;
      double x, y;
      if (std::fscanf(f, "%d %d %lf %lf", &c, &l, &x, &y) != 4) fail();
      if (c < 0 || c >= nc || l < 0 || l >= nl) fail();
      landmarks[l].obs.push_back({c, Scalar(x), Scalar(-y)});
    }
    for (auto& lm : landmarks) {
      std::sort(lm.obs.begin(), lm.obs.end(), [](const Observation& a, const Observation& b) { return a.cam < b.cam; });
      for (size_t i = 1; i < lm.obs.size(); ++i)
        if (lm.obs[i].cam == lm.obs[i - 1].cam) {
          std::fclose(f);
          throw std::runtime_error("Invalid file '" + path + "'");  // duplicate (camera, landmark)
        }
    }
    const detail::Mat3 flip = {1, 0, 0, 0, -1, 0, 0, 0, -1};
    for (int i = 0; i < nc; ++i) {
      double p[9];
      for (double& v : p)
        if (std::fscanf(f, "%lf", &v) != 1) fail();
      const detail::Mat3 R = detail::mul(flip, detail::so3_exp({p[0], p[1], p[2]}));
      double q[4];
      detail::rot_to_quat(R, q);
      auto& cam = cameras[i];
      for (int j = 0; j < 4; ++j) cam[j] = Scalar(q[j]);
      cam[4] = Scalar(p[3]);
      cam[5] = Scalar(-p[4]);
      cam[6] = Scalar(-p[5]);
      cam[7] = Scalar(p[6]);
      cam[8] = Scalar(p[7]);
      cam[9] = Scalar(p[8]);
    }
    for (int i = 0; i < nl; ++i) {
      double p[3];
      for (double& v : p)
        if (std::fscanf(f, "%lf", &v) != 1) fail();
      landmarks[i].p_w = {Scalar(p[0]), Scalar(p[1]), Scalar(p[2])};
    }
    std::fclose(f);
  }

  // X <- s (X - median), camera centres likewise; s = new_scale / MAD(L1)
  void normalize(double new_scale) {
    const size_t n = landmarks.size();
    std::vector<double> tmp(n);
    detail::Vec3 med;
    for (int j = 0; j < 3; ++j) {
      for (size_t i = 0; i < n; ++i) tmp[i] = landmarks[i].p_w[j];
      med[j] = detail::median_upper(tmp);
    }
    for (size_t i = 0; i < n; ++i) {
      const auto& p = landmarks[i].p_w;
      tmp[i] = std::abs(p[0] - med[0]) + std::abs(p[1] - med[1]) + std::abs(p[2] - med[2]);
    }
    const double scale = new_scale / detail::median_upper(tmp);
    for (auto& lm : landmarks)
      for (int j = 0; j < 3; ++j) lm.p_w[j] = Scalar(scale * (lm.p_w[j] - med[j]));
    for (auto& cam : cameras) {
      double q[4] = {double(cam[0]), double(cam[1]), double(cam[2]), double(cam[3])};
      const detail::Mat3 R = detail::quat_to_rot(q);
      detail::Vec3 c = detail::mul_t(R, {-double(cam[4]), -double(cam[5]), -double(cam[6])});  // camera centre
      for (int j = 0; j < 3; ++j) c[j] = scale * (c[j] - med[j]);
      const detail::Vec3 t = detail::mul(R, c);
      for (int j = 0; j < 3; ++j) cam[4 + j] = Scalar(-t[j]);
    }
  }

  // Gaussian noise on camera centres (world frame), camera rotations (local) and
  // points. Like the reference this uses std::default_random_engine, whose
  // stream is implementation defined.
  void perturb(double rotation_sigma, double translation_sigma, double landmark_sigma, int seed) {
    std::default_random_engine eng = seed < 0 ? std::default_random_engine{std::random_device{}()}
                                              : std::default_random_engine{static_cast<unsigned>(seed)};
    std::normal_distribution<double> normal;
    auto noise = [&](double sigma) { return detail::Vec3{normal(eng) * sigma, normal(eng) * sigma, normal(eng) * sigma}; };
    if (rotation_sigma > 0 || translation_sigma > 0) {
      for (auto& cam : cameras) {
        double q[4] = {double(cam[0]), double(cam[1]), double(cam[2]), double(cam[3])};
        detail::Mat3 R = detail::quat_to_rot(q);
        if (translation_sigma > 0) {
          detail::Vec3 c = detail::mul_t(R, {-double(cam[4]), -double(cam[5]), -double(cam[6])});
          const detail::Vec3 d = noise(translation_sigma);
          for (int j = 0; j < 3; ++j) c[j] += d[j];
          const detail::Vec3 t = detail::mul(R, c);
          for (int j = 0; j < 3; ++j) cam[4 + j] = Scalar(-t[j]);
        }
        if (rotation_sigma > 0) {
          R = detail::mul(detail::so3_exp(noise(rotation_sigma)), R);
          detail::rot_to_quat(R, q);
          for (int j = 0; j < 4; ++j) cam[j] = Scalar(q[j]);
        }
      }
    }
    if (landmark_sigma > 0)
      for (auto& lm : landmarks) {
        const detail::Vec3 d = noise(landmark_sigma);
        for (int j = 0; j < 3; ++j) lm.p_w[j] += Scalar(d[j]);
      }
  }

  // drop observations with depth < threshold, then landmarks with < 2 observations
  void filter_obs(double threshold) {
    if (threshold <= 0) return;
    for (auto& lm : landmarks) {
      auto bad = [&](const Observation& o) {
        const auto& cam = cameras[o.cam];
        double q[4] = {double(cam[0]), double(cam[1]), double(cam[2]), double(cam[3])};
        const detail::Mat3 R = detail::quat_to_rot(q);
        const double z = R[6] * lm.p_w[0] + R[7] * lm.p_w[1] + R[8] * lm.p_w[2] + cam[6];
        return z < threshold;
      };
      lm.obs.erase(std::remove_if(lm.obs.begin(), lm.obs.end(), bad), lm.obs.end());
    }
    landmarks.erase(std::remove_if(landmarks.begin(), landmarks.end(),
                                   [](const Landmark& l) { return l.obs.size() < 2; }),
                    landmarks.end());
  }

  template <class S2>
  BalProblem<S2> copy_cast() const {
    BalProblem<S2> out;
    out.cameras.resize(cameras.size());
    for (size_t i = 0; i < cameras.size(); ++i)
      for (int j = 0; j < 10; ++j) out.cameras[i][j] = S2(cameras[i][j]);
    out.landmarks.resize(landmarks.size());
    for (size_t i = 0; i < landmarks.size(); ++i) {
      for (int j = 0; j < 3; ++j) out.landmarks[i].p_w[j] = S2(landmarks[i].p_w[j]);
      for (const auto& o : landmarks[i].obs) out.landmarks[i].obs.push_back({o.cam, S2(o.x), S2(o.y)});
    }
    return out;
  }

  // flat views consumed by the C ABI
  void to_csr(std::vector<int64_t>& off, std::vector<int32_t>& cam, std::vector<Scalar>& xy) const {
    off.assign(1, 0);
    cam.clear();
    xy.clear();
    for (const auto& lm : landmarks) {
      for (const auto& o : lm.obs) {
        cam.push_back(o.cam);
        xy.push_back(o.x);
        xy.push_back(o.y);
      }
      off.push_back(int64_t(cam.size()));
    }
  }
  void copy_to_state(std::vector<Scalar>& cams, std::vector<Scalar>& lms) const {
    cams.resize(10 * cameras.size());
    lms.resize(3 * landmarks.size());
    for (size_t i = 0; i < cameras.size(); ++i) std::copy(cameras[i].begin(), cameras[i].end(), cams.begin() + 10 * i);
    for (size_t i = 0; i < landmarks.size(); ++i) std::copy(landmarks[i].p_w.begin(), landmarks[i].p_w.end(), lms.begin() + 3 * i);
  }
  void copy_from_state(const std::vector<Scalar>& cams, const std::vector<Scalar>& lms) {
    for (size_t i = 0; i < cameras.size(); ++i) std::copy(cams.begin() + 10 * i, cams.begin() + 10 * i + 10, cameras[i].begin());
    for (size_t i = 0; i < landmarks.size(); ++i) std::copy(lms.begin() + 3 * i, lms.begin() + 3 * i + 3, landmarks[i].p_w.begin());
  }
};

// load (double) -> normalize -> perturb -> filter -> cast
template <class Scalar>
BalProblem<Scalar> load_normalized_bal_problem(const BalDatasetOptions& o) {
  BalProblem<double> p;
  p.load_bal(o.input);
  if (o.normalize) p.normalize(o.normalization_scale);
  p.perturb(o.rotation_sigma, o.translation_sigma, o.point_sigma, o.random_seed);
  p.filter_obs(o.init_depth_threshold);
  return p.template copy_cast<Scalar>();
}

}  // namespace rootba_hip
