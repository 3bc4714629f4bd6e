// bal_problem.hpp — host-side BAL data model of the MI355X-native solver.
//
// Mirrors the parts of the reference's `rootba::BalProblem<Scalar>`
// (reference src/rootba/bal/bal_problem.hpp:61-234, bal_problem.cpp) that sit
// on either side of the accelerated path: BAL text loading (:190-282),
// normalisation (:428-469), perturbation (:508-554), depth filtering (:471-506),
// the flat camera state of `Camera::params()` (bal_problem.hpp:84-95) and the
// load pipeline order of `load_normalized_bal_problem` (:794-832).
// Own implementation on plain arrays (no Eigen/Sophus). One-off host work outside
// the accelerated hot path, but it dominates end-to-end time on the large
// problems, so the loader is a parallel tokeniser straight into the CSR/SoA
// layout of the C ABI.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cfloat>
#include <charconv>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <fstream>
#include <utility>
#include <vector>

namespace rootba_hip {

struct BalDatasetOptions {  // reference src/rootba/bal/bal_dataset_options.hpp:40-100
  enum class DatasetType { AUTO = 0, ROOTBA, BAL, BUNDLER };
  std::string input;
  DatasetType input_type = DatasetType::AUTO;
  bool save_output = false;
  std::string output_optimized_path = "optimized.cereal";
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
// number of worker threads for the one-off host passes (loader, filter)
inline int host_threads() {
  if (const char* e = std::getenv("RBA_HOST_THREADS")) return std::max(1, std::atoi(e));
  return int(std::max(1u, std::min(64u, std::thread::hardware_concurrency())));
}
template <class F>
inline void parallel_for(size_t n, int threads, F&& f) {  // f(begin, end, thread)
  threads = int(std::max<size_t>(1, std::min<size_t>(threads, n)));
  if (threads == 1) return f(size_t(0), n, 0);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t] { f(n * t / threads, n * (t + 1) / threads, t); });
  for (auto& th : pool) th.join();
}
inline bool is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r'; }

// Decimal text -> double, correctly rounded like fscanf("%lf"). Up to 15 significant
// digits and |exponent| <= 22 the value is m * 10^e with both factors exact in double
// (one rounding, Clinger's fast path). BAL files carry 17 digits ("%.16e"): up to 19
// digits and |exponent| <= 27 both factors are exact in the x87 64-bit mantissa, the
// product has one rounding there, and narrowing it to 53 bits is the correctly rounded
// result unless the 64-bit value sits next to a rounding midpoint — detected from its low
// 11 bits, about 3 in 2048 tokens — which, like everything else, goes through strtod.
// (libstdc++ 11's std::from_chars<double> wraps strtod in newlocale/uselocale, which
// takes a process-wide lock and does not scale over threads.)
// Returns the end of the token or nullptr.
inline const char* parse_double(const char* c, const char* stop, double& out) {
  static const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                    1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  const char* const begin = c;
  bool neg = false;
  if (c < stop && (*c == '-' || *c == '+')) neg = *c++ == '-';
  uint64_t m = 0;
  int sig = 0, exp10 = 0, ndig = 0;
  for (; c < stop && *c >= '0' && *c <= '9'; ++c, ++ndig) {
    if (sig > 0 || *c != '0') ++sig;
    if (sig <= 19) m = m * 10 + uint64_t(*c - '0'); else ++exp10;
  }
  if (c < stop && *c == '.') {
    for (++c; c < stop && *c >= '0' && *c <= '9'; ++c, ++ndig) {
      if (sig > 0 || *c != '0') ++sig;
      if (sig <= 19) { m = m * 10 + uint64_t(*c - '0'); --exp10; }
    }
  }
  if (ndig == 0) return nullptr;
  if (c < stop && (*c == 'e' || *c == 'E')) {
    const char* e = c + 1;
    bool eneg = false;
    if (e < stop && (*e == '-' || *e == '+')) eneg = *e++ == '-';
    int ev = 0, ed = 0;
    for (; e < stop && *e >= '0' && *e <= '9'; ++e, ++ed) ev = std::min(ev * 10 + (*e - '0'), 100000);
    if (ed == 0) return nullptr;
    exp10 += eneg ? -ev : ev;
    c = e;
  }
  if (sig <= 15 && exp10 >= -22 && exp10 <= 22) {
    const double v = exp10 < 0 ? double(m) / kPow10[-exp10] : double(m) * kPow10[exp10];
    out = neg ? -v : v;
    return c;
  }
#if LDBL_MANT_DIG == 64
  if (sig <= 19 && exp10 >= -27 && exp10 <= 27) {
    static const struct Pow10L {
      long double v[28];
      Pow10L() {
        v[0] = 1.0L;
        for (int i = 1; i < 28; ++i) v[i] = v[i - 1] * 10.0L;  // exact: 5^27 < 2^63
      }
    } kPow10L;
    const long double v = exp10 < 0 ? static_cast<long double>(m) / kPow10L.v[-exp10]
                                    : static_cast<long double>(m) * kPow10L.v[exp10];
    uint64_t mant;
    std::memcpy(&mant, &v, sizeof(mant));
    const unsigned low = unsigned(mant & 0x7FFu);
    if (low < 0x3FFu || low > 0x401u) {
      const double d = static_cast<double>(v);
      out = neg ? -d : d;
      return c;
    }
  }
#endif
  // slow path: strtod needs a terminated token
  const std::string tok(begin, c);
  char* endp = nullptr;
  out = std::strtod(tok.c_str(), &endp);
  return endp == tok.c_str() + tok.size() ? c : nullptr;
}

// read-only mapping of a whole file
struct MappedFile {
  const char* data = nullptr;
  size_t size = 0;
  explicit MappedFile(const std::string& path) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("Could not open '" + path + "'");
    struct stat st;
    if (::fstat(fd, &st) != 0 || st.st_size <= 0) {
      ::close(fd);
      throw std::runtime_error("Failed to parse '" + path + "'");
    }
    size = size_t(st.st_size);
    // MAP_POPULATE: one in-kernel pass instead of a page fault per 4 KB from every parser thread
    void* m = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) throw std::runtime_error("Could not map '" + path + "'");
    data = static_cast<const char*>(m);
  }
  ~MappedFile() {
    if (data) ::munmap(const_cast<char*>(data), size);
  }
  MappedFile(const MappedFile&) = delete;
  MappedFile& operator=(const MappedFile&) = delete;
};
}  // namespace detail

// Flat CSR / SoA storage — the layout `rba_create` takes (include/rootba_hip.h), so
// loading goes from the text file to the device without a per-landmark container
// (the reference keeps a std::map per landmark, bal_problem.hpp:117-127).
template <class Scalar>
class BalProblem {
 public:
  static constexpr int CAM_STATE_SIZE = 10;  // qx qy qz qw tx ty tz f k1 k2

  std::vector<std::array<Scalar, 10>> cameras;
  std::vector<std::array<Scalar, 3>> points;  // landmark positions p_w
  std::vector<int64_t> lm_off;                // [num_landmarks + 1] observation ranges
  std::vector<int32_t> obs_cam;               // ascending camera index inside a landmark (std::map order)
  std::vector<Scalar> obs_xy;                 // 2 per observation

  int num_cameras() const { return int(cameras.size()); }
  int num_landmarks() const { return int(points.size()); }
  int64_t num_observations() const { return int64_t(obs_cam.size()); }

  // BAL text format: header, observations (cam lm x y), 9 parameters per camera
  // (Rodrigues, t, f, k1, k2), 3 per landmark (reference bal_problem.cpp:190-282, which
  // reads it with fscanf into a map per landmark). The camera looks down -z with y up
  // in BAL; here +z forward / y down, so y and z axes are flipped on load.
  // Parallel two-pass tokeniser over the memory-mapped file: pass 1 counts the tokens
  // of each chunk, pass 2 converts them knowing their global index; observations are
  // then bucketed by landmark with a counting sort.
  void load_bal(const std::string& path, int threads = 0) {
    if (threads <= 0) threads = detail::host_threads();
    const detail::MappedFile file(path);
    const char* const base = file.data;
    const char* const end = base + file.size;
    auto fail = [&]() -> void { throw std::runtime_error("Failed to parse '" + path + "'"); };

    // header
    const char* cur = base;
    long long hdr[3];
    for (long long& h : hdr) {
      while (cur < end && detail::is_space(*cur)) ++cur;
      const auto r = std::from_chars(cur, end, h);
      if (r.ec != std::errc() || h <= 0) fail();
      cur = r.ptr;
    }
    const int64_t nc = hdr[0], nl = hdr[1], no = hdr[2];
    if (nc > INT32_MAX || nl > INT32_MAX) fail();
    const int64_t n_tokens = 4 * no + 9 * nc + 3 * nl;

    // chunk boundaries on whitespace
    const size_t body = size_t(end - cur);
    const int T = int(std::max<size_t>(1, std::min<size_t>(size_t(threads), body / 4096 + 1)));
    std::vector<const char*> cut(T + 1);
    cut[0] = cur;
    cut[T] = end;
    for (int t = 1; t < T; ++t) {
      const char* c = cur + body * t / T;
      while (c < end && !detail::is_space(*c)) ++c;
      cut[t] = c;
    }
    std::vector<int64_t> first_token(T + 1, 0);
    detail::parallel_for(size_t(T), T, [&](size_t b, size_t e, int) {
      for (size_t t = b; t < e; ++t) {
        int64_t n = 0;
        bool in_tok = false;
        for (const char* c = cut[t]; c < cut[t + 1]; ++c) {
          const bool sp = detail::is_space(*c);
          n += (!sp && !in_tok);
          in_tok = !sp;
        }
        first_token[t + 1] = n;
      }
    });
    for (int t = 0; t < T; ++t) first_token[t + 1] += first_token[t];
    if (first_token[T] < n_tokens) fail();  // trailing tokens are ignored like fscanf would

    std::vector<int32_t> raw_cam(no), raw_lm(no);
    std::vector<double> raw_xy(size_t(2) * no), params(size_t(9) * nc + size_t(3) * nl);
    std::atomic<bool> bad{false};
    detail::parallel_for(size_t(T), T, [&](size_t b, size_t e, int) {
      for (size_t t = b; t < e; ++t) {
        int64_t tok = first_token[t];
        const char* c = cut[t];
        const char* const stop = cut[t + 1];
        while (tok < n_tokens) {
          while (c < stop && detail::is_space(*c)) ++c;
          if (c >= stop) break;
          if (tok < 4 * no) {
            const int64_t o = tok >> 2;
            const int field = int(tok & 3);
            if (field < 2) {
              long long v = -1;
              const auto r = std::from_chars(c, stop, v);
              if (r.ec != std::errc() || v < 0 || v >= (field == 0 ? nc : nl)) return bad.store(true);
              (field == 0 ? raw_cam : raw_lm)[o] = int32_t(v);
              c = r.ptr;
            } else {
              double v;
              c = detail::parse_double(c, stop, v);
              if (!c) return bad.store(true);
              raw_xy[2 * o + (field - 2)] = field == 3 ? -v : v;
            }
          } else {
            double v;
            c = detail::parse_double(c, stop, v);
            if (!c) return bad.store(true);
            params[tok - 4 * no] = v;
          }
          if (c < stop && !detail::is_space(*c)) return bad.store(true);  // e.g. "1.5" where an index belongs
          ++tok;
        }
      }
    });
    if (bad.load()) fail();

    // counting sort by landmark (stable in file order), then camera order inside
    lm_off.assign(nl + 1, 0);
    for (int64_t o = 0; o < no; ++o) ++lm_off[raw_lm[o] + 1];
    for (int64_t l = 0; l < nl; ++l) lm_off[l + 1] += lm_off[l];
    obs_cam.resize(no);
    obs_xy.resize(size_t(2) * no);
    {
      std::vector<int64_t> fill(lm_off.begin(), lm_off.end() - 1);
      for (int64_t o = 0; o < no; ++o) {
        const int64_t d = fill[raw_lm[o]]++;
        obs_cam[d] = raw_cam[o];
        obs_xy[2 * d] = Scalar(raw_xy[2 * o]);
        obs_xy[2 * d + 1] = Scalar(raw_xy[2 * o + 1]);
      }
    }
    std::atomic<bool> dup{false};
    detail::parallel_for(size_t(nl), threads, [&](size_t b, size_t e, int) {
      std::vector<std::pair<int32_t, std::array<Scalar, 2>>> tmp;
      for (size_t l = b; l < e; ++l) {
        const int64_t o0 = lm_off[l], o1 = lm_off[l + 1];
        bool sorted = true;
        for (int64_t o = o0 + 1; o < o1; ++o) sorted = sorted && obs_cam[o - 1] < obs_cam[o];
        if (sorted) continue;
        tmp.clear();
        for (int64_t o = o0; o < o1; ++o) tmp.push_back({obs_cam[o], {obs_xy[2 * o], obs_xy[2 * o + 1]}});
        std::sort(tmp.begin(), tmp.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        for (int64_t o = o0; o < o1; ++o) {
          obs_cam[o] = tmp[o - o0].first;
          obs_xy[2 * o] = tmp[o - o0].second[0];
          obs_xy[2 * o + 1] = tmp[o - o0].second[1];
          if (o > o0 && obs_cam[o] == obs_cam[o - 1]) dup.store(true);
        }
      }
    });
    if (dup.load()) throw std::runtime_error("Invalid file '" + path + "'");  // duplicate (camera, landmark)

    const detail::Mat3 flip = {1, 0, 0, 0, -1, 0, 0, 0, -1};
    cameras.assign(nc, {});
    for (int64_t i = 0; i < nc; ++i) {
      const double* p = params.data() + 9 * i;
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
    points.resize(nl);
    const double* pp = params.data() + 9 * nc;
    for (int64_t i = 0; i < nl; ++i) points[i] = {Scalar(pp[3 * i]), Scalar(pp[3 * i + 1]), Scalar(pp[3 * i + 2])};
  }

  // Bundler "bundle.out" (reference BalProblem::load_bundler, bal_problem.cpp:284-404): one comment line ('#' first, up to the
  // end of the line), "num_cameras num_points", per camera 15 numbers (f k1 k2, R row-major 3x3, t), per point its position,
  // a colour (read and ignored) and a view list: count, then (camera, key, x, y) per view. Cameras with f == 0 are
  // uninitialised: dropped, later ones renumbered, their views skipped. Same axis convention as load_bal (camera y and z
  // axes inverted, image y inverted). A serial pass over the memory-mapped file (these files are small next to BAL dumps).
  void load_bundler(const std::string& path) {
    const detail::MappedFile file(path);
    const char* cur = file.data;
    const char* const end = file.data + file.size;
    auto fail = [&]() -> void { throw std::runtime_error("Failed to parse '" + path + "'"); };
    if (cur >= end || *cur != '#') fail();  // "non-comment line; expected comment..."
    while (cur < end && *cur != '\n') ++cur;
    if (cur >= end) fail();  // "could not read comment line"
    ++cur;
    auto next_int = [&]() {
      while (cur < end && detail::is_space(*cur)) ++cur;
      long long v = 0;
      const auto r = std::from_chars(cur, end, v);
      if (r.ec != std::errc() || (r.ptr < end && !detail::is_space(*r.ptr))) fail();
      cur = r.ptr;
      return v;
    };
    auto next_double = [&]() {
      while (cur < end && detail::is_space(*cur)) ++cur;
      double v = 0;
      const char* c = cur < end ? detail::parse_double(cur, end, v) : nullptr;
      if (!c || (c < end && !detail::is_space(*c))) fail();
      cur = c;
      return v;
    };
    const long long nc_file = next_int(), nl = next_int();
    if (nc_file <= 0 || nl <= 0 || nc_file > INT32_MAX || nl > INT32_MAX) fail();
    const detail::Mat3 flip = {1, 0, 0, 0, -1, 0, 0, 0, -1};
    std::vector<int32_t> remap(size_t(nc_file), -1);
    cameras.clear();
    for (long long i = 0; i < nc_file; ++i) {
      double p[15];
      for (double& v : p) v = next_double();
      if (p[0] == 0) continue;  // focal length 0: uninitialised camera
      remap[size_t(i)] = int32_t(cameras.size());
      const detail::Mat3 R = detail::mul(flip, detail::Mat3{p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11]});
      double q[4];
      detail::rot_to_quat(R, q);
      std::array<Scalar, 10> cam;
      for (int j = 0; j < 4; ++j) cam[j] = Scalar(q[j]);
      cam[4] = Scalar(p[12]);
      cam[5] = Scalar(-p[13]);
      cam[6] = Scalar(-p[14]);
      cam[7] = Scalar(p[0]);
      cam[8] = Scalar(p[1]);
      cam[9] = Scalar(p[2]);
      cameras.push_back(cam);
    }
    points.assign(size_t(nl), {});
    lm_off.assign(1, 0);
    obs_cam.clear();
    obs_xy.clear();
    std::vector<std::pair<int32_t, std::array<Scalar, 2>>> views;
    for (long long l = 0; l < nl; ++l) {
      for (int j = 0; j < 3; ++j) points[size_t(l)][j] = Scalar(next_double());
      for (int j = 0; j < 3; ++j) (void)next_double();  // colour
      const long long k = next_int();
      if (k < 0) fail();
      views.clear();
      for (long long v = 0; v < k; ++v) {
        const long long c = next_int();
        (void)next_int();  // key (feature) index
        const double x = next_double(), y = next_double();
        if (c < 0 || c >= nc_file) continue;     // (the reference's map lookup finds nothing: view skipped)
        if (remap[size_t(c)] < 0) continue;      // view of an uninitialised camera
        views.push_back({remap[size_t(c)], {Scalar(x), Scalar(-y)}});
      }
      std::stable_sort(views.begin(), views.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
      for (size_t v = 0; v < views.size(); ++v) {
        if (v > 0 && views[v].first == views[v - 1].first) throw std::runtime_error("Invalid file '" + path + "'");
        obs_cam.push_back(views[v].first);
        obs_xy.push_back(views[v].second[0]);
        obs_xy.push_back(views[v].second[1]);
      }
      lm_off.push_back(int64_t(obs_cam.size()));
    }
  }

  // X <- s (X - median), camera centres likewise; s = new_scale / MAD(L1)
  void normalize(double new_scale) {
    const size_t n = points.size();
    std::vector<double> tmp(n);
    detail::Vec3 med;
    for (int j = 0; j < 3; ++j) {
      for (size_t i = 0; i < n; ++i) tmp[i] = points[i][j];
      med[j] = detail::median_upper(tmp);
    }
    for (size_t i = 0; i < n; ++i) {
      const auto& p = points[i];
      tmp[i] = std::abs(p[0] - med[0]) + std::abs(p[1] - med[1]) + std::abs(p[2] - med[2]);
    }
    const double scale = new_scale / detail::median_upper(tmp);
    for (auto& p : points)
      for (int j = 0; j < 3; ++j) p[j] = Scalar(scale * (p[j] - med[j]));
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
    // A FRESH distribution object per 3-vector, as the reference's perturbation<T, N>() has (bal_problem.cpp:105-114):
    // libstdc++'s normal_distribution produces values in pairs and caches the second one, so a shared object would
    // consume the engine differently from the fourth draw on (the tests hold this pipeline to the reference's own loader).
    auto noise = [&](double sigma) {
      std::normal_distribution<double> normal;
      detail::Vec3 v;
      for (int j = 0; j < 3; ++j) v[j] = normal(eng) * sigma;
      return v;
    };
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
      for (auto& p : points) {
        const detail::Vec3 d = noise(landmark_sigma);
        for (int j = 0; j < 3; ++j) p[j] += Scalar(d[j]);
      }
  }

  // drop observations with depth < threshold, then landmarks with < 2 observations
  void filter_obs(double threshold, int threads = 0) {
    if (threshold <= 0) return;
    if (threads <= 0) threads = detail::host_threads();
    const int64_t nl = num_landmarks();
    std::vector<std::array<double, 4>> row2(cameras.size());  // third row of R and t_z
    for (size_t i = 0; i < cameras.size(); ++i) {
      const auto& cam = cameras[i];
      double q[4] = {double(cam[0]), double(cam[1]), double(cam[2]), double(cam[3])};
      const detail::Mat3 R = detail::quat_to_rot(q);
      row2[i] = {R[6], R[7], R[8], double(cam[6])};
    }
    std::vector<uint8_t> keep(obs_cam.size());
    std::vector<int64_t> kept(nl + 1, 0);
    detail::parallel_for(size_t(nl), threads, [&](size_t b, size_t e, int) {
      for (size_t l = b; l < e; ++l) {
        const auto& p = points[l];
        int64_t n = 0;
        for (int64_t o = lm_off[l]; o < lm_off[l + 1]; ++o) {
          const auto& r = row2[obs_cam[o]];
          keep[o] = !(r[0] * p[0] + r[1] * p[1] + r[2] * p[2] + r[3] < threshold);
          n += keep[o];
        }
        kept[l + 1] = n >= 2 ? n : 0;
      }
    });
    int64_t w_lm = 0, w_obs = 0;
    const std::vector<int64_t> old_off = lm_off;  // compaction below overwrites lm_off in place
    for (int64_t l = 0; l < nl; ++l) {
      const int64_t o0 = old_off[l], o1 = old_off[l + 1];
      if (kept[l + 1] == 0) continue;
      for (int64_t o = o0; o < o1; ++o)
        if (keep[o]) {
          obs_cam[w_obs] = obs_cam[o];
          obs_xy[2 * w_obs] = obs_xy[2 * o];
          obs_xy[2 * w_obs + 1] = obs_xy[2 * o + 1];
          ++w_obs;
        }
      points[w_lm] = points[l];
      lm_off[++w_lm] = w_obs;
    }
    points.resize(w_lm);
    lm_off.resize(w_lm + 1);
    obs_cam.resize(w_obs);
    obs_xy.resize(size_t(2) * w_obs);
  }

  // 1 - (non-zero 9x9 blocks of the reduced camera system) / n_c^2: cameras are coupled when they
  // observe a common landmark (reference bal_problem.cpp:647-712; byte mask instead of atomics)
  // ---- ".cereal" problem cache (reference BalProblem::save_rootba / load_rootba, bal_problem.cpp:137-181, 406-427) ------
  // A cereal BinaryOutputArchive of {file_info; cameras; landmarks}, always in double (the reference casts a float
  // problem to double before saving, :419-427). What fixes the byte layout:
  //   in the reference tree:  FileInfo = {type = "rootba::BalProblem", version = "1.0"} written first
  //                           (util/serialization.hpp:51-60, 150-155; bal_problem_io.hpp:48); BalProblem = cameras, landmarks;
  //                           Camera = T_c_w, intrinsics; Landmark = p_w, obs; Observation = pos (bal_problem_io.hpp:59-81);
  //   cereal's portable-binary conventions: arithmetic values raw little-endian, std::string / std::vector / std::map
  //                           preceded by a uint64 element count, a map entry = key then value, name-value wrappers add
  //                           nothing;
  //   basalt-headers' serialisers (un-vendored): SE3 as px py pz qx qy qz qw, fixed-size Eigen matrices as
  //                           raw coefficients, BalCamera as f k1 k2.
  // The last line cannot be checked against a file written by the reference here, so it is NOT taken on trust by the
  // reader: load_rootba parses under each plausible variant of it (translation-first or quaternion-first SE3; matrices
  // with or without an int32 rows/cols prefix, the intrinsics too or not) and accepts the one variant that consumes the
  // file exactly with unit quaternions, matrix shapes as declared, camera indices in range and strictly ascending inside
  // every landmark (std::map order). save_rootba writes the first variant.
  struct CerealLayout {
    bool translation_first = true;  // SE3: px py pz qx qy qz qw (else qx qy qz qw px py pz)
    bool matrix_dims = false;       // int32 rows, int32 cols before every Eigen matrix
    bool intrinsics_dims = false;   // ... also before the camera's parameter vector
  };
  static constexpr const char* kFileType = "rootba::BalProblem";
  static constexpr const char* kFileVersion = "1.0";

  bool save_rootba(const std::string& path, const CerealLayout& lay = CerealLayout()) const {
    std::string out;
    auto put = [&](const void* p, size_t n) { out.append(static_cast<const char*>(p), n); };
    auto put_u64 = [&](uint64_t v) { put(&v, 8); };
    auto put_f64 = [&](double v) { put(&v, 8); };
    auto put_str = [&](const std::string& v) {
      put_u64(v.size());
      put(v.data(), v.size());
    };
    auto put_dims = [&](bool on, int32_t r, int32_t c) {
      if (on) {
        put(&r, 4);
        put(&c, 4);
      }
    };
    put_str(kFileType);
    put_str(kFileVersion);
    put_u64(cameras.size());
    for (const auto& cam : cameras) {
      if (lay.translation_first) {
        for (int j = 4; j < 7; ++j) put_f64(double(cam[j]));
        for (int j = 0; j < 4; ++j) put_f64(double(cam[j]));
      } else {
        for (int j = 0; j < 7; ++j) put_f64(double(cam[j]));
      }
      put_dims(lay.matrix_dims && lay.intrinsics_dims, 3, 1);
      for (int j = 7; j < 10; ++j) put_f64(double(cam[j]));
    }
    put_u64(points.size());
    for (size_t l = 0; l < points.size(); ++l) {
      put_dims(lay.matrix_dims, 3, 1);
      for (int j = 0; j < 3; ++j) put_f64(double(points[l][j]));
      put_u64(uint64_t(lm_off[l + 1] - lm_off[l]));
      for (int64_t o = lm_off[l]; o < lm_off[l + 1]; ++o) {
        const int32_t c = obs_cam[o];
        put(&c, 4);
        put_dims(lay.matrix_dims, 2, 1);
        put_f64(double(obs_xy[2 * o]));
        put_f64(double(obs_xy[2 * o + 1]));
      }
    }
    std::ofstream os(path, std::ios::binary);
    if (!os.is_open()) return false;
    os.write(out.data(), std::streamsize(out.size()));
    return bool(os);
  }

  // returns false (and leaves *this untouched) when the file is not a BalProblem cache in any accepted layout;
  // `accepted`, if given, receives the layout that validated
  bool load_rootba(const std::string& path, std::string* error = nullptr, CerealLayout* accepted = nullptr) {
    const detail::MappedFile file(path);
    const char* const base = file.data;
    const size_t size = file.size;
    auto fail = [&](const std::string& why) {
      if (error) *error = why;
      return false;
    };
    size_t hdr = 0;
    auto get_str = [&](std::string& v) {
      uint64_t n;
      if (hdr + 8 > size) return false;
      std::memcpy(&n, base + hdr, 8);
      hdr += 8;
      if (n > size - hdr) return false;
      v.assign(base + hdr, size_t(n));
      hdr += size_t(n);
      return true;
    };
    std::string type, version;
    if (!get_str(type) || !get_str(version)) return fail("truncated file_info");
    if (type != kFileType) return fail("loaded file has different type '" + type + "'");
    if (version != kFileVersion) return fail("unknown version '" + version + "'");
    std::string last_why = "no layout matched";
    for (int v = 0; v < 6; ++v) {
      CerealLayout lay;
      lay.translation_first = (v & 1) == 0;
      lay.matrix_dims = v >= 2;
      lay.intrinsics_dims = v >= 4;
      BalProblem<double> p;
      size_t at = hdr;
      bool ok = true;
      std::string why;
      auto need = [&](size_t n) {
        if (ok && at + n > size) {
          ok = false;
          why = "truncated";
        }
        return ok;
      };
      auto get_u64 = [&]() {
        uint64_t x = 0;
        if (need(8)) {
          std::memcpy(&x, base + at, 8);
          at += 8;
        }
        return x;
      };
      auto get_f64 = [&]() {
        double x = 0;
        if (need(8)) {
          std::memcpy(&x, base + at, 8);
          at += 8;
        }
        return x;
      };
      auto get_i32 = [&]() {
        int32_t x = 0;
        if (need(4)) {
          std::memcpy(&x, base + at, 4);
          at += 4;
        }
        return x;
      };
      auto dims = [&](bool on, int32_t r, int32_t c) {
        if (!on) return;
        const int32_t rr = get_i32(), cc = get_i32();
        if (ok && (rr != r || cc != c)) {
          ok = false;
          why = "matrix shape";
        }
      };
      const uint64_t nc = get_u64();
      if (ok && (nc == 0 || nc > (size - at) / 80)) {
        ok = false;
        why = "camera count";
      }
      if (ok) p.cameras.resize(size_t(nc));
      for (uint64_t i = 0; ok && i < nc; ++i) {
        auto& cam = p.cameras[size_t(i)];
        if (lay.translation_first) {
          for (int j = 4; j < 7; ++j) cam[j] = get_f64();
          for (int j = 0; j < 4; ++j) cam[j] = get_f64();
        } else {
          for (int j = 0; j < 7; ++j) cam[j] = get_f64();
        }
        dims(lay.matrix_dims && lay.intrinsics_dims, 3, 1);
        for (int j = 7; j < 10; ++j) cam[j] = get_f64();
        const double n2 = cam[0] * cam[0] + cam[1] * cam[1] + cam[2] * cam[2] + cam[3] * cam[3];
        if (ok && !(std::abs(n2 - 1.0) < 1e-6)) {
          ok = false;
          why = "quaternion norm";
        }
      }
      const uint64_t nl = get_u64();
      if (ok && nl > (size - at) / 32) {
        ok = false;
        why = "landmark count";
      }
      if (ok) {
        p.points.resize(size_t(nl));
        p.lm_off.assign(1, 0);
        p.lm_off.reserve(size_t(nl) + 1);
      }
      for (uint64_t l = 0; ok && l < nl; ++l) {
        dims(lay.matrix_dims, 3, 1);
        for (int j = 0; j < 3; ++j) p.points[size_t(l)][j] = get_f64();
        const uint64_t k = get_u64();
        if (ok && k > (size - at) / 20) {
          ok = false;
          why = "observation count";
        }
        int32_t prev = -1;
        for (uint64_t o = 0; ok && o < k; ++o) {
          const int32_t c = get_i32();
          dims(lay.matrix_dims, 2, 1);
          const double x = get_f64(), y = get_f64();
          if (ok && (c <= prev || c >= int64_t(nc))) {
            ok = false;
            why = "camera index";
          }
          prev = c;
          p.obs_cam.push_back(c);
          p.obs_xy.push_back(x);
          p.obs_xy.push_back(y);
        }
        p.lm_off.push_back(int64_t(p.obs_cam.size()));
      }
      if (ok && at != size) {
        ok = false;
        why = "trailing bytes";
      }
      if (ok) {
        *this = p.template copy_cast<Scalar>();
        if (accepted) *accepted = lay;
        return true;
      }
      last_why = why;
    }
    return fail("not a BalProblem cache in any accepted layout (" + last_why + ")");
  }

  double compute_rcs_sparsity() const {
    const size_t nc = cameras.size();
    std::vector<uint8_t> mask(nc * nc, 0);
    for (int l = 0; l < num_landmarks(); ++l)
      for (int64_t i = lm_off[l]; i < lm_off[l + 1]; ++i)
        for (int64_t j = lm_off[l]; j < i; ++j) mask[size_t(obs_cam[i]) * nc + obs_cam[j]] = 1;  // cam_j < cam_i
    size_t lower = 0;
    for (uint8_t m : mask) lower += m;
    return 1.0 - double(nc + 2 * lower) / double(nc * nc);
  }

  template <class S2>
  BalProblem<S2> copy_cast() const {
    BalProblem<S2> out;
    out.cameras.resize(cameras.size());
    for (size_t i = 0; i < cameras.size(); ++i)
      for (int j = 0; j < 10; ++j) out.cameras[i][j] = S2(cameras[i][j]);
    out.points.resize(points.size());
    for (size_t i = 0; i < points.size(); ++i)
      for (int j = 0; j < 3; ++j) out.points[i][j] = S2(points[i][j]);
    out.lm_off = lm_off;
    out.obs_cam = obs_cam;
    out.obs_xy.assign(obs_xy.begin(), obs_xy.end());
    return out;
  }

  // flat views consumed by the C ABI
  void copy_to_state(std::vector<Scalar>& cams, std::vector<Scalar>& lms) const {
    cams.resize(10 * cameras.size());
    lms.resize(3 * points.size());
    for (size_t i = 0; i < cameras.size(); ++i) std::copy(cameras[i].begin(), cameras[i].end(), cams.begin() + 10 * i);
    for (size_t i = 0; i < points.size(); ++i) std::copy(points[i].begin(), points[i].end(), lms.begin() + 3 * i);
  }
  void copy_from_state(const std::vector<Scalar>& cams, const std::vector<Scalar>& lms) {
    for (size_t i = 0; i < cameras.size(); ++i) std::copy(cams.begin() + 10 * i, cams.begin() + 10 * i + 10, cameras[i].begin());
    for (size_t i = 0; i < points.size(); ++i) std::copy(lms.begin() + 3 * i, lms.begin() + 3 * i + 3, points[i].begin());
  }
};

// load (double) -> normalize -> perturb -> filter -> cast; optional wall times of the two halves
template <class Scalar>
BalProblem<Scalar> load_normalized_bal_problem(const BalDatasetOptions& o, double* load_seconds = nullptr,
                                               double* preprocess_seconds = nullptr) {
  const auto t0 = std::chrono::steady_clock::now();
  BalProblem<double> p;
  // autodetect_input_type (bal_problem.cpp:122-135), on the FILE NAME: "*.cereal" is the problem cache, a name containing
  // "bundle" a Bundler file, everything else BAL text
  BalDatasetOptions::DatasetType type = o.input_type;
  if (type == BalDatasetOptions::DatasetType::AUTO) {
    const size_t slash = o.input.find_last_of('/');
    const std::string name = slash == std::string::npos ? o.input : o.input.substr(slash + 1);
    const bool cereal = name.size() >= 7 && name.compare(name.size() - 7, 7, ".cereal") == 0;
    type = cereal ? BalDatasetOptions::DatasetType::ROOTBA
                  : name.find("bundle") != std::string::npos ? BalDatasetOptions::DatasetType::BUNDLER
                                                             : BalDatasetOptions::DatasetType::BAL;
  }
  if (type == BalDatasetOptions::DatasetType::ROOTBA) {
    std::string why;
    if (!p.load_rootba(o.input, &why)) throw std::runtime_error("Failed to load " + o.input + ": " + why);
  } else if (type == BalDatasetOptions::DatasetType::BUNDLER) {
    p.load_bundler(o.input);
  } else {
    p.load_bal(o.input);
  }
  const auto t1 = std::chrono::steady_clock::now();
  if (o.normalize) p.normalize(o.normalization_scale);
  p.perturb(o.rotation_sigma, o.translation_sigma, o.point_sigma, o.random_seed);
  p.filter_obs(o.init_depth_threshold);
  auto out = p.template copy_cast<Scalar>();
  const auto t2 = std::chrono::steady_clock::now();
  if (load_seconds) *load_seconds = std::chrono::duration<double>(t1 - t0).count();
  if (preprocess_seconds) *preprocess_seconds = std::chrono::duration<double>(t2 - t1).count();
  return out;
}

}  // namespace rootba_hip
