// solver.hip — host side of librootba_hip.so: device-resident solver state, the
// C ABI of include/rootba_hip.h, and the LM driver.
//
// Mirrors, call for call, what LinearizorQR does with LinearizationQR in the
// reference (src/rootba/solver/linearizor_qr.cpp:53-291) and what
// optimize_lm_ours does with the Linearizor
// (src/rootba/solver/bal_bundle_adjustment.cpp:249-544); everything indented
// under those calls runs on the GPU (kernels.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <limits>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rootba_hip.h"
#include "kernels.hpp"
#include "kernels_big.hpp"
#include "kernels_cam.hpp"
#include "kernels_s1.hpp"
#include "kernels_sc.hpp"
#include "kernels_pcg.hpp"
#include "kernels_pcgp.hpp"
#include "kernels_a64.hpp"

namespace {

thread_local std::string g_last_error;

struct HipError {
  std::string msg;
  int code;
};

#define HIP_CHECK(expr)                                                              \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      throw HipError{std::string(#expr) + ": " + hipGetErrorString(_e) + " (" +     \
                         __FILE__ + ":" + std::to_string(__LINE__) + ")",            \
                     RBA_ERR_HIP};                                                   \
    }                                                                                \
  } while (0)

// sets a flag for a scope (cleared on every exit path, exceptions included)
struct FlagScope {
  bool& flag;
  FlagScope(bool& f, bool on) : flag(f) { flag = on; }
  ~FlagScope() { flag = false; }
  FlagScope(const FlagScope&) = delete;
  FlagScope& operator=(const FlagScope&) = delete;
};

// one step of a host spin loop on a pinned word: lets the sibling hardware thread run
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}

template <class T>
class DevBuf {
 public:
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void alloc(size_t n) {
    release();
    n_ = n;
    if (n) HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p_), n * sizeof(T)));
  }
  void release() {
    if (p_) (void)hipFree(p_);
    p_ = nullptr;
    n_ = 0;
  }
  T* get() const { return p_; }
  size_t size() const { return n_; }
  void upload(const T* src, size_t n, hipStream_t st) {
    HIP_CHECK(hipMemcpyAsync(p_, src, n * sizeof(T), hipMemcpyHostToDevice, st));
  }
  void download(T* dst, size_t n, hipStream_t st) const {
    HIP_CHECK(hipMemcpyAsync(dst, p_, n * sizeof(T), hipMemcpyDeviceToHost, st));
  }
  void zero(hipStream_t st) {
    if (n_) HIP_CHECK(hipMemsetAsync(p_, 0, n_ * sizeof(T), st));
  }

 private:
  T* p_ = nullptr;
  size_t n_ = 0;
};

double wall_seconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch())
      .count();
}

// ---------------------------------------------------------------------------
// RCCL, loaded lazily (single-GPU runs never touch it)
// ---------------------------------------------------------------------------
struct Rccl {
  using UniqueId = struct { char internal[128]; };
  void* lib = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommAbort)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;  // (optional: reduce_ranges)
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load() {
    if (lib) return true;
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return false;
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(lib, "ncclCommAbort"));
    CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(lib, "ncclCommCount"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
    Reduce = reinterpret_cast<decltype(Reduce)>(dlsym(lib, "ncclReduce"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    return GetUniqueId && CommInitRank && CommDestroy && AllReduce;
  }
};
Rccl g_rccl;
// ncclDataType_t / ncclRedOp_t values (rccl.h): ncclInt32 = 2, ncclFloat32 = 7,
// ncclFloat64 = 8; ncclSum = 0, ncclMax = 2
constexpr int kNcclInt32 = 2, kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclSum = 0, kNcclMax = 2;

}  // namespace

// ---------------------------------------------------------------------------
// Solver
// ---------------------------------------------------------------------------
struct rba_solver {
  virtual ~rba_solver() = default;
  virtual void comm_init(int rank, int nranks, const void* uid) = 0;
  virtual void comm_init_callback(int rank, int nranks, rba_allreduce_fn fn, void* ctx) = 0;
  virtual void comm_info(int* rank, int* nranks, int* transport) = 0;
  virtual void comm_stats(int64_t* calls, int64_t* bytes, double* seconds) = 0;
  // called from ANOTHER thread when a sibling rank of a sharded handle has failed: this rank's pending and future
  // collectives must fail instead of waiting for the missing rank
  virtual void comm_abort() {}
  virtual void set_state(const void* cams, const void* lms) = 0;
  virtual void get_state(void* cams, void* lms) = 0;
  virtual void backup() = 0;
  virtual void restore() = 0;
  virtual void compute_error(rba_residual_info* out) = 0;
  virtual int linearize(void* jp_diag2_out) = 0;
  virtual int solve(double lambda, void* inc_out, rba_cg_summary* cg) = 0;
  virtual int stage2(double lambda, void* b_out, void* blocks_out) = 0;
  virtual void right_multiply(const void* x, void* y) = 0;
  virtual void right_multiply_explicit(const void* x, void* y) = 0;
  virtual int apply(const void* inc, double* l_diff, bool update_cams) = 0;
  virtual int optimize_lm(rba_lm_iteration* log, int max_rows, int* n_rows, int* term) = 0;
  virtual void lm_begin() = 0;
  virtual int lm_step(rba_lm_iteration* out) = 0;
  virtual int lm_termination() const = 0;
  virtual void device_sync() = 0;
  virtual int64_t debug_read_A(int vec) = 0;
  virtual void get_timings(rba_iter_timings* out) = 0;
  virtual void get_substage_timings(rba_substage_timings* out) = 0;
  virtual void get_jl_col_scale(void* out) = 0;
  virtual void get_pose_scaling(void* out) = 0;
  virtual void get_landmark_R(int damped, void* R6, void* q3) = 0;
  virtual void get_landmark_q2tr_norm(void* out) = 0;
  virtual void get_problem_stats(int64_t* storage, int64_t* hx_bytes, int64_t* hx_flops) = 0;
  virtual void get_byte_model(rba_byte_model* out) = 0;
  virtual void get_pcg_counters(rba_pcg_counters* out) = 0;
  virtual void get_reduced_matrix_info(rba_reduced_matrix_info* out) = 0;
};

namespace {

constexpr int kNumClasses = 5;
constexpr int kClassCH[kNumClasses] = {1, 2, 4, 8, 16};

template <class S>
class Solver final : public rba_solver {
 public:
  Solver(int device, int n_cams, int n_lms, const int64_t* lm_off, const int32_t* obs_cam,
         const S* obs_xy, const rba_options& opt, const double* obs_xy64 = nullptr)
      : device_(device), n_cams_(n_cams), n_lms_(n_lms), opt_(opt), mixed_(obs_xy64 != nullptr) {
    // input validation first: nothing to release if it throws
    for (int l = 0; l < n_lms; ++l) {
      const int64_t k = lm_off[l + 1] - lm_off[l];
      if (k < 2)
        throw HipError{"every landmark needs >= 2 observations (landmark " + std::to_string(l) + ")",
                       RBA_ERR_INVALID_ARGUMENT};
      for (int64_t q = lm_off[l]; q < lm_off[l + 1]; ++q) {
        if (obs_cam[q] < 0 || obs_cam[q] >= n_cams)
          throw HipError{"camera index out of range", RBA_ERR_INVALID_ARGUMENT};
        if (q > lm_off[l] && obs_cam[q] <= obs_cam[q - 1])
          throw HipError{"camera indices must be strictly ascending inside a landmark",
                         RBA_ERR_INVALID_ARGUMENT};
      }
    }
    try {
      construct(lm_off, obs_cam, obs_xy, obs_xy64);
    } catch (...) {
      release_resources();  // the destructor does not run for a throwing constructor
      throw;
    }
  }

  void construct(const int64_t* lm_off, const int32_t* obs_cam, const S* obs_xy, const double* obs_xy64) {
    const int n_cams = n_cams_, n_lms = n_lms_;
    HIP_CHECK(hipSetDevice(device_));
    HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithFlags(&side_stream_, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
    n_obs_ = lm_off[n_lms];
    nvec_ = 9 * n_cams_;
    sc_ = opt_.solver_type == 1;
    read_debug_env();
    {
      int dev = 0, cus = 0;
      HIP_CHECK(hipGetDevice(&dev));
      HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      if (cus > 0) n_cus_ = cus;
    }

    // ---- sort landmarks by number of observations (stable) ----------------
    perm_.resize(n_lms);
    std::iota(perm_.begin(), perm_.end(), 0);
    // (With device-scope atomics a secondary sort by first camera is a loss: neighbouring lanes then
    //  scatter-add to the same addresses and the atomics serialise - dense blocks 538 -> 660 us, one tile per
    //  wave from the factors 238 -> 397 us on venice - so the dense-block configuration keeps the input order.)
    // Secondary key: first camera. With the LDS-private products (k_hx_implicit_lds) neighbouring lanes adding to
    // the same cameras cost nothing, the camera windows of large problems need this locality, and the
    // camera-major gathers get slightly faster (venice: stage 1 440 -> 410 us). RBA_SORT_BY_CAMERA=0 keeps the
    // input order inside a track length.
    int sort_by_camera = 1;
    if (sort_by_camera) {
      // More cameras than an LDS window holds: the sort only pays if tracks are local in camera index. Without
      // that locality (e.g. an unordered photo collection) the windows would miss, the products would fall back to
      // device atomics, and those are fastest on the INPUT order - so it is kept then.
      const int win = int(kHxLdsMaxBytes / (9 * sizeof(double)));
      if (n_cams > win) {
        int64_t local_rows = 0, rows = 0;
        for (int l = 0; l < n_lms; ++l) {
          const int64_t k = lm_off[l + 1] - lm_off[l];
          rows += 2 * k;
          if (obs_cam[lm_off[l + 1] - 1] - obs_cam[lm_off[l]] < win / 2) local_rows += 2 * k;
        }
        if (double(local_rows) < 0.9 * double(rows)) sort_by_camera = 0;
      }
    }
    if (env_.sort_by_camera >= 0) sort_by_camera = env_.sort_by_camera;
    // track-length classes: the common refinement of the wave-tile classes (k <= 2, 4, 8, 16, 32, 64, 112) and
    // of the chunk classes of the matrix-free E0 products (k <= 7, 14, 28, 56, 112): every class range stays contiguous
    auto k_class = [](int64_t k) {
      static const int bounds[] = {2, 4, 7, 8, 14, 16, 28, 32, 56, 64, 112};
      int c = 0;
      for (int b : bounds) c += k > b ? 1 : 0;
      return c;
    };
    {
      // one 64-bit key per landmark (a comparator that recomputes it costs seconds at final-13682 size):
      // unsorted: k; sorted: class | first camera | k (equal tracks next to each other)
      std::vector<uint64_t> key(n_lms);
      for (int l = 0; l < n_lms; ++l) {
        const int64_t k = lm_off[l + 1] - lm_off[l];
        key[l] = sort_by_camera ? (uint64_t(k_class(k)) << 56) | (uint64_t(uint32_t(obs_cam[lm_off[l]])) << 24) |
                                      uint64_t(std::min<int64_t>(k, (1 << 24) - 1))
                                : uint64_t(k);
      }
      std::stable_sort(perm_.begin(), perm_.end(), [&](int a, int b) { return key[a] < key[b]; });
    }
    std::vector<int> lm_k(n_lms);
    std::vector<int64_t> lm_obs(n_lms + 1);
    std::vector<int> s_obs_cam(n_obs_), s_obs_lm(n_obs_);
    std::vector<S> s_obs_xy(2 * size_t(n_obs_));
    std::vector<double> s_obs_xy64(mixed_ ? 2 * size_t(n_obs_) : 0);
    int64_t o = 0;
    int kmax = 0;
    hx_bytes_ = 0;
    hx_flops_ = 0;
    storage_bytes_ = 0;
    for (int s = 0; s < n_lms; ++s) {
      const int l = perm_[s];
      const int k = int(lm_off[l + 1] - lm_off[l]);
      lm_k[s] = k;
      lm_obs[s] = o;
      kmax = std::max(kmax, k);
      for (int i = 0; i < k; ++i) {
        const int64_t src = lm_off[l] + i;
        s_obs_cam[o] = obs_cam[src];
        s_obs_lm[o] = s;
        s_obs_xy[2 * o] = obs_xy[2 * src];
        s_obs_xy[2 * o + 1] = obs_xy[2 * src + 1];
        if (mixed_) {
          s_obs_xy64[2 * o] = obs_xy64[2 * src];
          s_obs_xy64[2 * o + 1] = obs_xy64[2 * src + 1];
        }
        ++o;
      }
      // algorithmic counts in the reference's padded layout
      const int64_t pad = (4 - (9 * k) % 4) % 4;
      hx_bytes_ += int64_t(sizeof(S)) * 2 * k * (9 * k + pad);
      hx_flops_ += int64_t(72) * k * k;
      storage_bytes_ += int64_t(sizeof(S)) * (2 * k + 3) * (9 * k + pad + 4);
      // implicit-Q variant, formula of SURVEY.md §8d: s * (2k (9+3) + 3 (2k+3) + 12)
      hx_implicit_bytes_ += int64_t(sizeof(S)) * (2 * k * 12 + 3 * (2 * k + 3) + 12);
    }
    lm_obs[n_lms] = o;
    // CSC index camera -> observations (sorted-observation numbering), used by
    // the camera-major reductions
    std::vector<int64_t> cam_off(n_cams + 1, 0);
    for (int64_t q = 0; q < n_obs_; ++q) ++cam_off[s_obs_cam[q] + 1];
    for (int c = 0; c < n_cams; ++c) cam_off[c + 1] += cam_off[c];
    std::vector<int> cam_obs(n_obs_);
    {
      std::vector<int64_t> cur(cam_off.begin(), cam_off.end() - 1);
      for (int64_t q = 0; q < n_obs_; ++q) cam_obs[cur[s_obs_cam[q]]++] = int(q);
    }
    hx_bytes_ += 4 * n_obs_ + int64_t(sizeof(S)) * 2 * 9 * n_cams_;
    hx_implicit_bytes_ += 4 * n_obs_ + int64_t(sizeof(S)) * 2 * 9 * n_cams_;
    // class ranges (k <= 7*CH)
    int begin = 0;
    for (int c = 0; c < kNumClasses; ++c) {
      int end = begin;
      while (end < n_lms && lm_k[end] <= 7 * kClassCH[c]) ++end;
      cls_begin_[c] = begin;
      cls_end_[c] = end;
      begin = end;
    }
    // ranges of the implicit-Q operator: a landmark's 2k rows fit an aligned group of
    // P2 = 4, 8, 16, 32, 64 lanes, or 2 / 4 x 64 lanes
    {
      const int kmax_of[kNumImplicit] = {2, 4, 8, 16, 32, 64, 112};
      int b0 = 0;
      for (int c = 0; c < kNumImplicit; ++c) {
        int e0 = b0;
        while (e0 < n_lms && lm_k[e0] <= kmax_of[c]) ++e0;
        imp_begin_[c] = b0;
        imp_end_[c] = e0;
        b0 = e0;
      }
    }
    // wave tiles of the implicit-Q operator for k <= 32 (classes 0..4)
    std::vector<int> lm_tile(n_lms, -1), lm_lane0(n_lms, 0), tile_cam, tile_row;
    {
      const int p2_of[5] = {4, 8, 16, 32, 64};
      int tiles = 0;
      for (int c = 0; c < 5; ++c) {
        imp_tile_begin_[c] = tiles;
        const int lpw = 64 / p2_of[c];
        const int n = imp_end_[c] - imp_begin_[c];
        imp_tiles_[c] = (n + lpw - 1) / lpw;
        for (int q = 0; q < n; ++q) {
          lm_tile[imp_begin_[c] + q] = tiles + q / lpw;
          lm_lane0[imp_begin_[c] + q] = (q % lpw) * p2_of[c];
        }
        tiles += imp_tiles_[c];
      }
      n_tiles_ = tiles;
      tile_cam.assign(size_t(n_tiles_) * 64, -1);
      tile_row.assign(size_t(n_tiles_) * 64, -1);
      if (2 * n_obs_ > int64_t(std::numeric_limits<int>::max()))
        throw HipError{"more than 2^30 observations: block-row indices exceed 32 bits", RBA_ERR_UNSUPPORTED};
      for (int s2 = 0; s2 < n_lms; ++s2)
          if (lm_tile[s2] >= 0)
            for (int rr = 0; rr < 2 * lm_k[s2]; ++rr) {
              tile_cam[size_t(lm_tile[s2]) * 64 + lm_lane0[s2] + rr] = s_obs_cam[lm_obs[s2] + rr / 2];
              tile_row[size_t(lm_tile[s2]) * 64 + lm_lane0[s2] + rr] = int(2 * lm_obs[s2] + rr);
            }
    }
    // persistent workgroups of k_hx_implicit_lds (kernels.hpp: HxChunk): camera ranges x tile runs
    if (n_tiles_ > 0) {
      const int G = std::max(1, std::min(n_cus_, (n_tiles_ + 15) / 16));
      hx_win_ = std::min(n_cams, int(kHxLdsMaxBytes / (9 * sizeof(double))));
      if (env_.hx_win > 0) hx_win_ = std::max(1, std::min(hx_win_, env_.hx_win));  // tests
      auto first_cam = [&](int T) { return tile_cam[size_t(T) * 64]; };
      // runs of ascending first camera
      std::vector<int> run_begin{0};
      for (int T = 1; T < n_tiles_; ++T)
        if (first_cam(T) < first_cam(T - 1)) run_begin.push_back(T);
      run_begin.push_back(n_tiles_);
      const int n_runs = int(run_begin.size()) - 1;
      std::vector<rba::HxChunk> chunks(G);
      const bool windows = hx_win_ < n_cams;
      if (!windows || n_runs > rba::kHxMaxRuns) {
        // every camera fits (or the order has no camera locality): even split of the tiles, window from 0
        for (int g = 0; g < G; ++g) {
          rba::HxChunk c{};
          c.cam_lo = 0;
          c.n_ranges = 1;
          c.tile_begin[0] = int(int64_t(n_tiles_) * g / G);
          c.tile_end[0] = int(int64_t(n_tiles_) * (g + 1) / G);
          chunks[g] = c;
        }
      } else {
        // camera boundaries with equal tile counts, then per run the tiles whose first camera is in range
        std::vector<int64_t> cnt(size_t(n_cams) + 1, 0);
        for (int T = 0; T < n_tiles_; ++T) ++cnt[size_t(first_cam(T)) + 1];
        for (int c = 0; c < n_cams; ++c) cnt[size_t(c) + 1] += cnt[c];
        std::vector<int> bound(size_t(G) + 1, n_cams);
        bound[0] = 0;
        for (int g = 1; g < G; ++g) {
          const int64_t target = int64_t(n_tiles_) * g / G;
          bound[g] = int(std::lower_bound(cnt.begin(), cnt.end(), target) - cnt.begin());
          bound[g] = std::max(bound[g - 1], std::min(bound[g], n_cams));
        }
        for (int g = 0; g < G; ++g) {
          rba::HxChunk c{};
          c.cam_lo = std::max(0, std::min(bound[g], n_cams - hx_win_));
          c.n_ranges = 0;
          for (int r = 0; r < n_runs; ++r) {
            auto lb = [&](int cam) {  // first tile of run r with first camera >= cam
              int lo = run_begin[r], hi = run_begin[r + 1];
              while (lo < hi) {
                const int mid = (lo + hi) / 2;
                if (first_cam(mid) < cam) lo = mid + 1; else hi = mid;
              }
              return lo;
            };
            const int tb = lb(bound[g]), te = lb(bound[g + 1]);
            if (te > tb) {
              c.tile_begin[c.n_ranges] = tb;
              c.tile_end[c.n_ranges] = te;
              ++c.n_ranges;
            }
          }
          chunks[g] = c;
        }
      }
      // share of the block rows that land inside their workgroup's window
      int64_t covered = 0, total = 0;
      for (const auto& c : chunks)
        for (int r = 0; r < c.n_ranges; ++r)
          for (size_t q = size_t(c.tile_begin[r]) * 64; q < size_t(c.tile_end[r]) * 64; ++q)
            if (tile_cam[q] >= 0) {
              ++total;
              covered += unsigned(tile_cam[q] - c.cam_lo) < unsigned(hx_win_) ? 1 : 0;
            }
      hx_coverage_ = total > 0 ? double(covered) / double(total) : 1.0;
      d_hx_chunks_.alloc(chunks.size());
      d_hx_chunks_.upload(chunks.data(), chunks.size(), stream_);
      HIP_CHECK(hipStreamSynchronize(stream_));  // `chunks` is a local
      n_hx_chunks_ = G;
      if (env_.verbose)
        std::fprintf(stderr, "[rootba_hip] H*x: %d persistent workgroups, %d tile runs, window %d of %d cameras, %.2f %% of "
                     "the block rows inside their window\n", G, n_runs, hx_win_, n_cams, 100.0 * hx_coverage_);
    }
    // everything beyond k = 112: one workgroup per landmark (kernels_big.hpp)
    big_begin_ = begin;
    n_big_ = n_lms - begin;
    big_kmax_ = n_big_ > 0 ? kmax : 0;
    // ---- device memory ------------------------------------------------------
    d_lm_k_.alloc(n_lms);
    d_lm_obs_.alloc(n_lms + 1);
    d_obs_cam_.alloc(n_obs_);
    d_obs_lm_.alloc(n_obs_);
    d_obs_xy_.alloc(2 * size_t(n_obs_));
    d_lm_k_.upload(lm_k.data(), n_lms, stream_);
    d_lm_obs_.upload(lm_obs.data(), n_lms + 1, stream_);
    d_obs_cam_.upload(s_obs_cam.data(), n_obs_, stream_);
    d_obs_lm_.upload(s_obs_lm.data(), n_obs_, stream_);
    d_obs_xy_.upload(s_obs_xy.data(), 2 * size_t(n_obs_), stream_);
    d_cam_off_.alloc(n_cams + 1);
    d_cam_obs_.alloc(n_obs_);
    d_cam_off_.upload(cam_off.data(), n_cams + 1, stream_);
    d_cam_obs_.upload(cam_obs.data(), n_obs_, stream_);
    d_cams_.alloc(10 * size_t(n_cams));
    d_lms_.alloc(3 * size_t(n_lms));
    d_cams_bak_.alloc(10 * size_t(n_cams));
    d_lms_bak_.alloc(3 * size_t(n_lms));
    if (mixed_) {
      // RBA_MIXED: double master state + double observations for the cost; float everywhere else
      if (sc_) throw HipError{"RBA_MIXED is implemented for the SQUARE_ROOT solver", RBA_ERR_UNSUPPORTED};
      d_obs_xy64_.alloc(2 * size_t(n_obs_));
      d_obs_xy64_.upload(s_obs_xy64.data(), 2 * size_t(n_obs_), stream_);
      d_cams64_.alloc(10 * size_t(n_cams));
      d_lms64_.alloc(3 * size_t(n_lms));
      d_cams64_bak_.alloc(10 * size_t(n_cams));
      d_lms64_bak_.alloc(3 * size_t(n_lms));
      d_lm_inc_.alloc(3 * size_t(n_lms));
      HIP_CHECK(hipStreamSynchronize(stream_));  // s_obs_xy64 is a local
    }
    // LinearizorSC offers SCHUR_JACOBI and POWER_SCHUR_COMPLEMENT (linearizor_sc.cpp:158-176; JACOBI: LOG(FATAL))
    if (sc_ && opt_.preconditioner_type == 0)
      throw HipError{"SCHUR_COMPLEMENT solver: the SCHUR_JACOBI and POWER_SCHUR_COMPLEMENT preconditioners are "
                     "implemented (the reference's LinearizorSC has no JACOBI either)", RBA_ERR_UNSUPPORTED};
    // the dense landmark blocks and the QR by-products exist only for the square-root solver
    const size_t qr_obs = sc_ ? 0 : size_t(n_obs_);
    if (!sc_ && n_big_ > 0) {
      // global scratch of the long-track kernels: 8 scalars per block row (kernels_big.hpp)
      std::vector<int64_t> off(size_t(n_big_) + 1, 0);
      for (int q = 0; q < n_big_; ++q) off[q + 1] = off[q] + 2 * int64_t(lm_k[big_begin_ + q]);
      d_big_off_.alloc(off.size());
      d_big_off_.upload(off.data(), off.size(), stream_);
      d_big_scratch_.alloc(size_t(8) * off[n_big_]);
      HIP_CHECK(hipStreamSynchronize(stream_));
    }
    d_topd_.alloc(rba::kTd * qr_obs);
    if (env_.deterministic && !sc_) {
      // the row entries / landmark sums of the deterministic products (kernels.hpp: k_hx_det_gather, k_e0_det_gather)
      d_hx_u_.alloc(2 * size_t(n_obs_));
      d_e0_w_.alloc(3 * size_t(n_lms_));
    }
    d_JpS_.alloc(18 * size_t(n_obs_));
    d_JlS_.alloc(6 * qr_obs);
    d_rS_.alloc(2 * qr_obs);
    d_bsO_.alloc(5 * qr_obs);
    d_givens_.alloc(sc_ ? 0 : 16 * size_t(n_lms));
    d_Vh_.alloc(8 * qr_obs);
    if (sc_) {
      h_lm_obs_ = lm_obs;
      h_obs_cam_ = s_obs_cam;
      ex_nb_ = co_observing_cameras(lm_obs, s_obs_cam);
      build_sc_structure();
      d_sc_JlS_.alloc(6 * size_t(n_obs_));
      d_sc_rS_.alloc(2 * size_t(n_obs_));
      d_sc_M_.alloc(6 * size_t(n_lms_));
      d_sc_v_.alloc(3 * size_t(n_lms_));
      d_sc_Hinv_.alloc(9 * size_t(n_lms_));
      d_sc_hb_.alloc(3 * size_t(n_lms_));
      d_sc_W_.alloc(27 * size_t(n_obs_));
      d_sc_T_.alloc(27 * size_t(n_obs_));
      d_sc_bO_.alloc(9 * size_t(n_obs_));
    }
    // square-root solver: explicit reduced matrix for long PCG solves (see pcg())
    explicit_after_ = opt_.explicit_after;
    if (env_.explicit_after != INT_MIN) explicit_after_ = env_.explicit_after;
    if (explicit_after_ < 0) {  // auto: start with 6, then the measured break-even (see solve())
      explicit_auto_ = true;
      explicit_after_ = 6;
    }
    // (needs the SCHUR_JACOBI blocks of stage 2 as its diagonal)
    // The assembly gathers over per-block lists of observation pairs: sum_l k_l (k_l - 1) / 2 pairs of
    // 8 bytes (plus the 81-scalar blocks). Heavy-tailed track lengths make that O(sum k^2); it is
    // bounded here: above the budget (RBA_EX_PAIR_BUDGET_GB, default 24 GB of the 288) the solver stays
    // matrix-free, which needs no such lists.
    int64_t n_pairs_total = 0;
    for (int l = 0; l < n_lms; ++l) n_pairs_total += int64_t(lm_k[l]) * (lm_k[l] - 1) / 2;
    ex_pair_bytes_ = 8 * n_pairs_total;
    const double pair_budget_gb = env_.pair_budget_gb;
    const bool pairs_fit = double(ex_pair_bytes_) <= pair_budget_gb * 1e9;
    if (!pairs_fit && env_.verbose)
      std::fprintf(stderr, "[rootba_hip] pair lists of the reduced matrix would take %.1f GB (> %.1f GB): "
                           "products stay matrix-free\n", ex_pair_bytes_ * 1e-9, pair_budget_gb);
    if (!sc_ && explicit_after_ > 0 && pairs_fit) {
      h_lm_obs_ = lm_obs;
      h_obs_cam_ = s_obs_cam;
      ex_nb_ = co_observing_cameras(lm_obs, s_obs_cam);
      build_explicit_structure();
      // only the first explicit_after products of a solve are matrix-free: sample them densely
      if (env_.hx_timing_stride < 0) hx_timing_stride_ = 2;
    }
    d_tauH_.alloc(3 * size_t(n_lms));
    d_Zd_.alloc(9 * size_t(n_lms));
    d_Zd_.zero(stream_);
    d_LQ_.alloc(sc_ ? 0 : 12 * size_t(n_lms));
    if (n_tiles_ > 0) {
      // the tile map, per OBSERVATION slot (lanes 2 q and 2 q + 1 of a tile are the two rows of one observation)
      std::vector<int2> tile_obs(size_t(n_tiles_) * 32);
      for (size_t i = 0; i < tile_obs.size(); ++i) {
        if ((tile_cam[2 * i] >= 0) != (tile_cam[2 * i + 1] >= 0) || (tile_cam[2 * i] >= 0 &&
            (tile_cam[2 * i + 1] != tile_cam[2 * i] || tile_row[2 * i + 1] != tile_row[2 * i] + 1 || (tile_row[2 * i] & 1))))
          throw HipError{"tile map: the two lanes of an observation slot disagree", RBA_ERR_INVALID_ARGUMENT};
        tile_obs[i] = int2{tile_cam[2 * i], tile_cam[2 * i] >= 0 ? tile_row[2 * i] : -1};
      }
      d_OT_.alloc(tile_obs.size());
      d_OT_.upload(tile_obs.data(), tile_obs.size(), stream_);
    }
    n_obs_small_ = lm_obs[big_begin_];  // observations of the landmarks with k <= 112 (sorted first)
    n_obs_tiled_ = lm_obs[imp_end_[4]];  // ... with k <= 32 (the wave tiles)
    d_R0_.alloc(6 * size_t(n_lms));
    d_Rd_.alloc(6 * size_t(n_lms));
    d_q1trd_.alloc(3 * size_t(n_lms));
    d_damp_r_.alloc(3 * size_t(n_lms));
    d_jl_scale_.alloc(3 * size_t(n_lms));
    d_jp_diag2_.alloc(nvec_);
    d_pose_scaling_.alloc(nvec_);
    d_mid_.alloc(size_t(81) * n_cams);  // B_mid
    d_bb_.alloc(size_t(171) * n_cams);  // [b | blocks | diagonal blocks of the reduced matrix (JACOBI / series)]
    d_inv_.alloc(size_t(81) * n_cams);
    d_fail_.alloc(1);
    d_lm_ldiff_.alloc(n_lms);
    d_partials_.alloc(size_t(kReduceBlocks) * 8 + 16);
    d_partials_side_.alloc(size_t(kReduceBlocks) * 8 + 16);
    d_endred_.alloc(rba::kEndRed);
    d_endred_.zero(stream_);
    d_cg_.alloc(1);
    d_pcg_partials_.alloc(3 * rba::kPcgBlocks);
    for (auto* v : {&d_x_, &d_r_, &d_p_, &d_z_, &d_q_, &d_tmp_, &d_inc_, &d_vin_, &d_pw_t_, &d_pw_e_})
      v->alloc(nvec_);
    d_topd_.zero(stream_);
    d_pose_scaling_.zero(stream_);
    d_fail_.zero(stream_);
    d_partials_.zero(stream_);
    d_p2_.alloc(nvec_);
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_pinned_), 4096));
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_vec_stage_), size_t(nvec_) * sizeof(S)));
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_progress_), 64));
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_stamps_), kMaxStamps * sizeof(unsigned long long)));
    {
      int khz = 0;  // (the clock of wall_clock64() / s_memrealtime: 100 MHz on gfx950)
      if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device_) == hipSuccess && khz > 0)
        stamp_hz_ = double(khz) * 1e3;
      else
        (void)hipGetLastError();
    }
    h_progress_[0] = h_progress_[1] = 0;
    HIP_CHECK(hipEventCreate(&ev_asm0_));
    HIP_CHECK(hipEventCreate(&ev_asm1_));
    d_scratch_int_.alloc(1);
    hx_events_.assign(2 * kMaxHxEvents, nullptr);
    hx_event_call_.resize(kMaxHxEvents);
    for (auto& e : hx_events_) HIP_CHECK(hipEventCreate(&e));
    HIP_CHECK(hipStreamSynchronize(stream_));

    prm_.n_cams = n_cams;
    prm_.n_lms = n_lms;
    prm_.lm_k = d_lm_k_.get();
    prm_.lm_obs = d_lm_obs_.get();
    prm_.obs_cam = d_obs_cam_.get();
    prm_.obs_lm = d_obs_lm_.get();
    prm_.obs_xy = d_obs_xy_.get();
    prm_.cam_obs_off = d_cam_off_.get();
    prm_.cam_obs = d_cam_obs_.get();
    prm_.JpS = d_JpS_.get();
    prm_.JpT = d_JpS_.get() + 16 * size_t(n_obs_);  // (split storage of the rows: kernels.hpp, jp_row)
    prm_.JlS = d_JlS_.get();
    prm_.rS = d_rS_.get();
    prm_.bsO = d_bsO_.get();
    prm_.givens = d_givens_.get();
    prm_.Vh = d_Vh_.get();
    prm_.tauH = d_tauH_.get();
    prm_.Zd = d_Zd_.get();
    prm_.LQ = d_LQ_.get();
    prm_.OT = d_OT_.get();
    prm_.cams = d_cams_.get();
    prm_.lms = d_lms_.get();
    prm_.lm_inc = mixed_ ? d_lm_inc_.get() : nullptr;
    if (!sc_) {
      // the eight stage-2 coefficients themselves: only for the observations of the two-kernel back-substitution
      prm_.w8_begin = n_tiles_ > 0 ? n_obs_tiled_ : 0;
      d_W8_.alloc(size_t(8) * (n_obs_ - prm_.w8_begin));
      d_WA_.alloc(size_t(rba::kRecW) * n_obs_);
      d_xs_.alloc(nvec_);
    }
    prm_.W8 = d_W8_.get();
    prm_.WA = d_WA_.get();
    prm_.topd = d_topd_.get();
    prm_.R0 = d_R0_.get();
    prm_.Rd = d_Rd_.get();
    prm_.q1trd = d_q1trd_.get();
    prm_.damp_r = d_damp_r_.get();
    prm_.jl_scale = d_jl_scale_.get();
    prm_.jp_diag2 = d_jp_diag2_.get();
    prm_.pose_scaling = d_pose_scaling_.get();
    prm_.B_mid = d_mid_.get();
    prm_.b = d_bb_.get();
    prm_.blocks = d_bb_.get() + nvec_;
    prm_.fail_flag = d_fail_.get();
    prm_.lm_ldiff = d_lm_ldiff_.get();
    prm_.robust_norm = opt_.robust_norm;
    prm_.valid_only = opt_.use_valid_projections_only;
    // JACOBI and the power-series preconditioner both start from Hpp = sum Jp^T Jp
    prm_.jacobi = opt_.preconditioner_type == 0 || opt_.preconditioner_type == 2;
    prm_.sdiag = d_bb_.get() + size_t(90) * n_cams;
    prm_.want_sdiag = (prm_.jacobi && ex_ready_) ? 1 : 0;
    prm_.huber = S(opt_.huber_parameter);
    prm_.eps = opt_.jacobi_scaling_eps > 0 ? S(opt_.jacobi_scaling_eps) : rba::Eps<S>::eps_sqrt;
    if (mixed_) {
      // what k_compute_error<double> reads: topology, double observations, the double master state
      prm64_ = rba::Params<double>{};
      prm64_.n_cams = n_cams;
      prm64_.n_lms = n_lms;
      prm64_.obs_cam = prm_.obs_cam;
      prm64_.obs_lm = prm_.obs_lm;
      prm64_.obs_xy = d_obs_xy64_.get();
      prm64_.cams = d_cams64_.get();
      prm64_.lms = d_lms64_.get();
      prm64_.robust_norm = opt_.robust_norm;
      prm64_.valid_only = opt_.use_valid_projections_only;
      prm64_.huber = opt_.huber_parameter;
    }
    if (sc_) {
      scp_.n_cams = n_cams;
      scp_.n_lms = n_lms;
      scp_.n_obs = n_obs_;
      scp_.lm_k = prm_.lm_k;
      scp_.lm_obs = prm_.lm_obs;
      scp_.obs_cam = prm_.obs_cam;
      scp_.obs_lm = prm_.obs_lm;
      scp_.obs_xy = prm_.obs_xy;
      scp_.cam_obs_off = prm_.cam_obs_off;
      scp_.cam_obs = prm_.cam_obs;
      scp_.cams = prm_.cams;
      scp_.lms = prm_.lms;
      scp_.pose_scaling = prm_.pose_scaling;
      scp_.JpS = prm_.JpS;
      scp_.JlS = d_sc_JlS_.get();
      scp_.rS = d_sc_rS_.get();
      scp_.M = d_sc_M_.get();
      scp_.v = d_sc_v_.get();
      scp_.scale = prm_.jl_scale;
      scp_.Hinv = d_sc_Hinv_.get();
      scp_.hb = d_sc_hb_.get();
      scp_.W = d_sc_W_.get();
      scp_.T = d_sc_T_.get();
      scp_.bO = d_sc_bO_.get();
      scp_.b = prm_.b;  // (row_ptr, cols, diag_slot, vals: build_sc_structure)
      scp_.blocks = prm_.blocks;
      scp_.fail_flag = prm_.fail_flag;
      scp_.lm_ldiff = prm_.lm_ldiff;
      scp_.robust_norm = prm_.robust_norm;
      scp_.valid_only = prm_.valid_only;
      scp_.huber = prm_.huber;
      scp_.eps = prm_.eps;
    }
    // capture the launch graphs of the fused PCG now (one-off cost, not part of an LM iteration)
    if (n_items_ > 0 && (sc_ || ex_ready_)) build_pcg_graphs();
  }

  // nb[c] = the cameras that observe a landmark together with camera c (ascending, without c): the block structure of
  // the reduced camera matrix. Camera by camera over its landmarks with one row of marks - sum_l k_l^2 steps like the
  // pair lists, O(n_c + blocks) memory (rounds 1-3 kept dense n_c x n_c tables and capped the camera count at 20000).
  std::vector<std::vector<int>> co_observing_cameras(const std::vector<int64_t>& lm_obs,
                                                     const std::vector<int>& s_obs_cam) const {
    const size_t nc = size_t(n_cams_);
    std::vector<int64_t> cam_ptr(nc + 1, 0);
    for (int l = 0; l < n_lms_; ++l)
      for (int64_t o = lm_obs[l]; o < lm_obs[l + 1]; ++o) ++cam_ptr[size_t(s_obs_cam[o]) + 1];
    for (size_t c = 0; c < nc; ++c) cam_ptr[c + 1] += cam_ptr[c];
    std::vector<int> cam_lm(static_cast<size_t>(cam_ptr[nc]), 0);
    {
      std::vector<int64_t> fill(cam_ptr.begin(), cam_ptr.end() - 1);
      for (int l = 0; l < n_lms_; ++l)
        for (int64_t o = lm_obs[l]; o < lm_obs[l + 1]; ++o) cam_lm[size_t(fill[s_obs_cam[o]]++)] = l;
    }
    std::vector<std::vector<int>> nb(nc);
    std::vector<uint8_t> mark(nc, 0);
    for (size_t c = 0; c < nc; ++c) {
      std::vector<int>& list = nb[c];
      mark[c] = 1;
      for (int64_t q = cam_ptr[c]; q < cam_ptr[c + 1]; ++q) {
        const int l = cam_lm[size_t(q)];
        for (int64_t o = lm_obs[l]; o < lm_obs[l + 1]; ++o) {
          const int d = s_obs_cam[o];
          if (!mark[d]) {
            mark[d] = 1;
            list.push_back(d);
          }
        }
      }
      std::sort(list.begin(), list.end());
      mark[c] = 0;
      for (int d : list) mark[d] = 0;
    }
    return nb;
  }

  // Block-CSR structure for the explicit reduced matrix of the square-root solver, from the neighbour lists (the union
  // over ranks when landmarks are sharded), and the per-block lists of the LOCAL observation pairs (i < j) that
  // contribute to each strictly upper block
  void build_explicit_structure() {
    // HALF storage (kernels_pcg.hpp): every off-diagonal block {c, d} lives in the row of its OWNER, as the owner sees
    // it; the product's contribution to the OTHER row travels through a 9-double slot, and the slots a row receives
    // are contiguous. Everything here works on the sorted neighbour lists of co_observing_cameras().
    const size_t nc = size_t(n_cams_);
    constexpr int CB = rba::spmv_chunk_blocks<double>();
    const std::vector<std::vector<int>>& nb = ex_nb_;  // neighbours, ascending
    std::vector<std::vector<uint8_t>> own(nc);         // own[c][k] = 1: row c owns {c, nb[c][k]}
    std::vector<int> cnt(nc, 1);                       // blocks row c owns (incl. the diagonal one)
    for (size_t c = 0; c < nc; ++c) own[c].assign(nb[c].size(), 0);
    auto find = [&](int c, int d) {  // position of d in nb[c]
      return int(std::lower_bound(nb[c].begin(), nb[c].end(), d) - nb[c].begin());
    };
    // Ownership. Start: parity of c + d (every row owns about half of its blocks). Then blocks are handed over from
    // rows that own more than one wavefront's worth (CB blocks incl. the diagonal one: such a row needs a second
    // work item) to neighbours with room, directly or through a full neighbour (chains of two) - venice-1778: 2308
    // -> ~1890 work items for 55 K blocks, against 1792 wavefronts the chip holds at once. Deterministic: every rank
    // of a sharded run derives the same table from the same (united) structure.
    for (size_t c = 0; c < nc; ++c)
      for (size_t k = 0; k < nb[c].size(); ++k) {
        const size_t d = size_t(nb[c][k]);
        if (((c + d) & 1) == 0 ? c < d : c > d) {
          own[c][k] = 1;
          ++cnt[c];
        }
      }
    auto hand_over = [&](int from, int k_from, int to) {  // {from, to}: from -> to
      own[from][k_from] = 0;
      own[to][find(to, from)] = 1;
      --cnt[from];
      ++cnt[to];
    };
    for (int sweep = 0; sweep < 8; ++sweep) {
      int moved = 0;
      for (size_t o = 0; o < nc; ++o) {
        while (cnt[o] > CB) {
          bool done = false;
          for (size_t k = 0; k < nb[o].size() && !done; ++k)
            if (own[o][k] && cnt[nb[o][k]] < CB) {
              hand_over(int(o), int(k), nb[o][k]);
              done = true;
            }
          for (size_t q = 0; !done && q < nb[o].size(); ++q) {
            const int t = nb[o][q];
            if (!own[o][q] || cnt[t] != CB) continue;
            for (size_t k2 = 0; k2 < nb[t].size(); ++k2) {
              const int u = nb[t][k2];
              if (u != int(o) && own[t][k2] && cnt[u] < CB) {
                hand_over(t, int(k2), u);
                hand_over(int(o), int(q), t);
                done = true;
                break;
              }
            }
          }
          if (!done) break;
          ++moved;
        }
      }
      if (!moved) break;
    }
    // rows: the diagonal block and the owned ones, ascending column; slot_nb[c][k] = slot of {c, nb[c][k]} in row c (-1)
    std::vector<int> row_ptr(nc + 1, 0), cols, diag(nc), upper_slot, mirror_slot;
    std::vector<std::vector<int>> slot_nb(nc);
    int nnz = 0;
    for (size_t c = 0; c < nc; ++c) {
      row_ptr[c] = nnz;
      slot_nb[c].assign(nb[c].size(), -1);
      bool diag_done = false;
      for (size_t k = 0; k <= nb[c].size(); ++k) {
        if (!diag_done && (k == nb[c].size() || nb[c][k] > int(c))) {
          diag[c] = nnz++;
          cols.push_back(int(c));
          diag_done = true;
        }
        if (k < nb[c].size() && own[c][k]) {
          slot_nb[c][k] = nnz++;
          cols.push_back(nb[c][k]);
        }
      }
    }
    row_ptr[nc] = nnz;
    // slots of the transposed products, grouped by receiving row (ascending sender): tdst[slot of (c, d)] = position in
    // row d's run. A HEAVY row (more than kHalfLowerMax slots: dense co-visibility, e.g. a landmark seen by most
    // cameras) is not gathered by the work-items that consume q - one of them would walk hundreds of slots - but
    // summed by a wavefront of its own right behind the product (k_pcgs_reduce_slots) into one more "further item" of
    // that row: its run is hidden from the consumers (low_ptr: empty) and listed in heavy_rows.
    std::vector<int> low_ptr(2 * nc, 0), tdst(size_t(nnz), -1);
    std::vector<rba::HeavyRow> heavy_rows;
    {
      int n_slots = 0;
      for (size_t d = 0; d < nc; ++d) {
        const int begin = n_slots;
        for (size_t k = 0; k < nb[d].size(); ++k)
          if (!own[d][k]) {  // {c, d} lives in row c = nb[d][k]
            const int c = nb[d][k];
            tdst[slot_nb[c][find(c, int(d))]] = n_slots++;
          }
        const bool heavy = n_slots - begin > env_.half_lower_max;
        if (heavy) heavy_rows.push_back(rba::HeavyRow{int(d), begin, n_slots, -1});
        low_ptr[2 * d] = begin;
        low_ptr[2 * d + 1] = heavy ? begin : n_slots;
      }
      d_low_ptr_.alloc(low_ptr.size());
      d_low_ptr_.upload(low_ptr.data(), low_ptr.size(), stream_);
      d_tdst_.alloc(tdst.size());
      d_tdst_.upload(tdst.data(), tdst.size(), stream_);
      d_tpart_.alloc(size_t(9) * std::max(1, n_slots));
      d_tpart_.zero(stream_);
    }
    // the blocks as the assembly produces them: (c, d) with c < d from the observation pairs (i, j > i) of a landmark;
    // written straight into row c and / or transposed into row d, wherever it is stored. up_nb[c][k] = index of the
    // block {c, nb[c][k] > c}
    std::vector<std::vector<int>> up_nb(nc);
    for (size_t c = 0; c < nc; ++c) {
      up_nb[c].assign(nb[c].size(), -1);
      for (size_t k = 0; k < nb[c].size(); ++k) {
        const int d = nb[c][k];
        if (d < int(c)) continue;
        up_nb[c][k] = int(upper_slot.size());
        upper_slot.push_back(slot_nb[c][k]);                 // -1: stored in row d only
        mirror_slot.push_back(slot_nb[d][find(d, int(c))]);  // -1: stored in row c only
      }
    }
    const int n_upper = int(upper_slot.size());
    std::vector<int64_t> pair_ptr(size_t(n_upper) + 1, 0);
    // (cameras ascend inside a landmark, and so do the neighbours of a camera: for a fixed i the blocks (i, j > i) are
    //  found by one forward walk of nb[cam_i])
    auto for_each_pair = [&](auto&& f) {
      for (int l = 0; l < n_lms_; ++l) {
        const int64_t o0 = h_lm_obs_[l];
        const int k = int(h_lm_obs_[l + 1] - o0);
        for (int i = 0; i < k; ++i) {
          const int ci = h_obs_cam_[o0 + i];
          const std::vector<int>& ni = nb[ci];
          size_t pos = 0;
          for (int j = i + 1; j < k; ++j) {
            const int cj = h_obs_cam_[o0 + j];
            while (ni[pos] < cj) ++pos;  // (cj is a neighbour of ci: the lists come from these very pairs)
            f(up_nb[ci][pos], int(o0 + i), int(o0 + j));
          }
        }
      }
    };
    for_each_pair([&](int u, int, int) { ++pair_ptr[size_t(u) + 1]; });
    for (int t = 0; t < n_upper; ++t) pair_ptr[t + 1] += pair_ptr[t];
    const int64_t n_pairs = pair_ptr[n_upper];
    ex_pairs_ = n_pairs;
    std::vector<int> pair_oi(n_pairs), pair_oj(n_pairs);
    {
      std::vector<int64_t> fill(pair_ptr.begin(), pair_ptr.end() - 1);
      for_each_pair([&](int u, int oi, int oj) {
        const int64_t d = fill[u]++;
        pair_oi[d] = oi;
        pair_oj[d] = oj;
      });
    }
    ex_nnz_ = nnz;
    ex_n_upper_ = n_upper;
    // one chunk of 33 double blocks per item: every wavefront is one pass of loads (kernels_pcg.hpp)
    build_spmv_items(row_ptr, rba::spmv_chunk_blocks<double>(), &heavy_rows);
    build_pcgp_structure(nb, slot_nb, diag);
    n_heavy_ = int(heavy_rows.size());
    d_heavy_.alloc(std::max<size_t>(1, heavy_rows.size()));
    if (n_heavy_ > 0) d_heavy_.upload(heavy_rows.data(), heavy_rows.size(), stream_);
    d_ex_rowptr_.alloc(row_ptr.size());
    d_ex_cols_.alloc(cols.size());
    d_ex_diag_.alloc(diag.size());
    d_ex_upper_.alloc(upper_slot.size());
    d_ex_mirror_.alloc(mirror_slot.size());
    d_ex_pair_ptr_.alloc(pair_ptr.size());
    d_ex_pair_oi_.alloc(n_pairs);
    d_ex_pair_oj_.alloc(n_pairs);
    d_ex_vals_.alloc(size_t(81) * nnz + 4);  // + 4: the SpMV's last 16-byte load may run past the end
    // power-series preconditioner of a float solver: its terms run through a FLOAT copy of the matrix (series_f32())
    if (series_f32()) d_ex_vals32_.alloc(size_t(81) * nnz + 8);
    if (kA64) {
      // float solver: the matrix is assembled in double from the float factors (kernels_a64.hpp)
      d_a64_lq_.alloc(size_t(rba::kA64Lq) * n_lms_);
      d_a64_A_.alloc(size_t(4) * n_obs_);
      d_a64_rec_.alloc(size_t(rba::kA64Rec) * n_obs_);
    }
    d_ex_rowptr_.upload(row_ptr.data(), row_ptr.size(), stream_);
    d_ex_cols_.upload(cols.data(), cols.size(), stream_);
    d_ex_diag_.upload(diag.data(), diag.size(), stream_);
    d_ex_upper_.upload(upper_slot.data(), upper_slot.size(), stream_);
    d_ex_mirror_.upload(mirror_slot.data(), mirror_slot.size(), stream_);
    d_ex_pair_ptr_.upload(pair_ptr.data(), pair_ptr.size(), stream_);
    d_ex_pair_oi_.upload(pair_oi.data(), n_pairs, stream_);
    d_ex_pair_oj_.upload(pair_oj.data(), n_pairs, stream_);
    HIP_CHECK(hipStreamSynchronize(stream_));
    exp_ = rba::ScParams<S>{};
    exp_.n_cams = n_cams_;
    exp_.row_ptr = d_ex_rowptr_.get();
    exp_.cols = d_ex_cols_.get();
    exp_.vals = nullptr;  // the values are double for either solver scalar: with_matrix()
    ex_ready_ = true;
  }

  // Persistent PCG (kernels_pcgp.hpp): consecutive block rows in FULL storage per workgroup, one block per lane, rows
  // padded to whole quads of lanes; a workgroup holds at most 512 lanes and 56 rows. Not every matrix fits the chip's
  // register files (n_cus_ workgroups): `pg_ready_` says whether this one does. Deterministic from the (united)
  // structure: every rank of a sharded run derives the same tables.
  // half storage of the square-root solver's assembled matrix: a block lives in the row of its owner
  void build_pcgp_structure(const std::vector<std::vector<int>>& nb, const std::vector<std::vector<int>>& slot_nb,
                            const std::vector<int>& diag) {
    build_pcgp_structure(nb, [&](int c, int k) {  // 2 * slot + transposed of the block {c, nb[c][k]}; k < 0: the diagonal one
      if (k < 0) return 2 * diag[c];
      if (slot_nb[c][k] >= 0) return 2 * slot_nb[c][k];  // stored in row c as row c sees it
      const int d = nb[c][k];
      const int kk = int(std::lower_bound(nb[d].begin(), nb[d].end(), c) - nb[d].begin());
      return 2 * slot_nb[d][kk] + 1;  // stored in row d: S_cd = S_dc^T
    });
  }
  template <class SrcOf>
  void build_pcgp_structure(const std::vector<std::vector<int>>& nb, SrcOf&& src_of) {
    pg_ready_ = false;
    pg_G_ = 0;
    if (env_.pcg_persistent == 0) return;
    const int nc = n_cams_;
    constexpr int T = rba::kPgThreads;
    auto padded = [&](int c) { return (int(nb[c].size()) + 1 + 3) & ~3; };
    std::vector<rba::PgWorkgroup> wgs;
    {
      int c = 0;
      while (c < nc) {
        rba::PgWorkgroup w{c, 0, 0, 0};
        int lanes = 0;
        while (c < nc && w.nrows < rba::kPgMaxRows && lanes + padded(c) <= T) {
          lanes += padded(c);
          ++w.nrows;
          ++c;
        }
        if (w.nrows == 0) return;  // a row with more than 512 blocks: two-launch path
        wgs.push_back(w);
      }
    }
    const int G = int(wgs.size());
    if (G > std::min(n_cus_, rba::kPgMaxGroups)) {
      if (env_.verbose)
        std::fprintf(stderr, "[rootba_hip] persistent PCG: the matrix needs %d workgroups, the chip holds %d - two-launch path\n",
                     G, n_cus_);
      return;
    }
    std::vector<int> lane_src(size_t(G) * T, -1), stage_col(size_t(G) * T, -1), row_info(size_t(3) * nc, 0);
    std::vector<unsigned short> lane_col(size_t(G) * T, 0);
    std::vector<int> where(size_t(nc), -1);  // staged index of a camera in the current workgroup
    for (int g = 0; g < G; ++g) {
      rba::PgWorkgroup& w = wgs[g];
      // the distinct columns of the workgroup's rows, ascending
      std::vector<int> cols;
      for (int c = w.row0; c < w.row0 + w.nrows; ++c) {
        cols.push_back(c);
        cols.insert(cols.end(), nb[c].begin(), nb[c].end());
      }
      std::sort(cols.begin(), cols.end());
      cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
      w.ncols = int(cols.size());  // (<= the workgroup's lanes <= 512)
      for (int t = 0; t < w.ncols; ++t) {
        stage_col[size_t(g) * T + t] = cols[t];
        where[cols[t]] = t;
      }
      int lane = 0;
      for (int c = w.row0; c < w.row0 + w.nrows; ++c) {
        row_info[3 * c] = lane / 4;
        row_info[3 * c + 1] = padded(c) / 4;
        row_info[3 * c + 2] = where[c];
        bool diag_done = false;
        auto put = [&](int col, int src) {
          lane_src[size_t(g) * T + lane] = src;
          lane_col[size_t(g) * T + lane] = static_cast<unsigned short>(where[col]);
          ++lane;
        };
        for (size_t k = 0; k <= nb[c].size(); ++k) {
          if (!diag_done && (k == nb[c].size() || nb[c][k] > c)) {
            put(c, src_of(c, -1));
            diag_done = true;
          }
          if (k == nb[c].size()) break;
          put(nb[c][k], src_of(c, int(k)));
        }
        lane = (lane + 3) & ~3;
      }
      for (int t = 0; t < w.ncols; ++t) where[cols[t]] = -1;
    }
    d_pg_wg_.alloc(wgs.size());
    d_pg_lane_src_.alloc(lane_src.size());
    d_pg_lane_col_.alloc(lane_col.size());
    d_pg_stage_col_.alloc(stage_col.size());
    d_pg_row_info_.alloc(row_info.size());
    d_pg_wg_.upload(wgs.data(), wgs.size(), stream_);
    d_pg_lane_src_.upload(lane_src.data(), lane_src.size(), stream_);
    d_pg_lane_col_.upload(lane_col.data(), lane_col.size(), stream_);
    d_pg_stage_col_.upload(stage_col.data(), stage_col.size(), stream_);
    d_pg_row_info_.upload(row_info.data(), row_info.size(), stream_);
    constexpr int NR = rba::pg_vec_records<S>();
    d_pg_zg_.alloc(size_t(NR) * n_cams_);
    d_pg_xg_.alloc(size_t(NR) * n_cams_);
    d_pg_tg_.alloc(size_t(2) * NR * n_cams_);
    d_pg_part_.alloc(size_t(3) * rba::kPgReplicas * G);
    d_pg_zg_.zero(stream_);
    d_pg_xg_.zero(stream_);
    d_pg_tg_.zero(stream_);
    d_pg_part_.zero(stream_);
    HIP_CHECK(hipStreamSynchronize(stream_));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rba::k_pcgp<S>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  int(rba::pgp_lds_bytes<S>())));
    pg_G_ = G;
    pg_epoch_ = 1;
    pg_ready_ = true;
    if (env_.verbose)
      std::fprintf(stderr, "[rootba_hip] persistent PCG: %d workgroups of %d lanes for %d blocks in full storage\n", G, T,
                   sc_ ? sc_nnz_ : 2 * ex_nnz_ - nc);
  }

  // work items of the fused PCG's SpMV (kernels_pcg.hpp): one wavefront per block row, rows with
  // more than 64 * kSpmvChunksPerItem blocks are split (their partial sums are added in item order)
  void build_spmv_items(const std::vector<int>& row_ptr, int span, std::vector<rba::HeavyRow>* heavy) {
    destroy_pcg_graphs();  // they hold the addresses of the buffers (re)allocated here
    std::vector<rba::SpmvItem> items;
    std::vector<int> extra_ptr(size_t(n_cams_) + 1, 0);
    int n_extra = 0;
    size_t hi = 0;
    for (int c = 0; c < n_cams_; ++c) {
      extra_ptr[c] = n_extra;
      items.push_back(rba::SpmvItem{c, row_ptr[c], std::min(row_ptr[c] + span, row_ptr[c + 1]), -1});
      for (int s0 = row_ptr[c] + span; s0 < row_ptr[c + 1]; s0 += span)
        items.push_back(rba::SpmvItem{c, s0, std::min(s0 + span, row_ptr[c + 1]), n_extra++});
      // (half storage: the sum of a heavy row's received slots is one more "further item" of the row)
      if (heavy && hi < heavy->size() && (*heavy)[hi].row == c) (*heavy)[hi++].extra = n_extra++;
    }
    extra_ptr[n_cams_] = n_extra;
    n_items_ = int(items.size());
    d_items_.alloc(items.size());
    d_item_ptr_.alloc(extra_ptr.size());
    d_items_.upload(items.data(), items.size(), stream_);
    d_item_ptr_.upload(extra_ptr.data(), extra_ptr.size(), stream_);
    d_qpart_.alloc(size_t(9) * std::max(1, n_extra));
    d_qmain_.alloc(nvec_);
    d_pcgs_pq_.alloc(n_items_);
    HIP_CHECK(hipStreamSynchronize(stream_));
    static_assert(rba::spmv_lds_bytes<double>() <= 48 * 1024, "the SpMV staging buffer needs the large-LDS attribute");
  }

  // S = sum_l A_l^T A_l of the CURRENT damped blocks (valid until the next stage 2)
  void assemble_explicit() {
    const bool measure = explicit_auto_ && !asm_measured_ && !asm_pending_;
    if (measure) HIP_CHECK(hipEventRecord(ev_asm0_, stream_));
    if (comm_ || cb_fn_) d_ex_vals_.zero(stream_);  // sharded: blocks without local pairs must be 0
    ++pcg_counters_.assemblies;
    assemble_values();
    if (d_ex_vals32_.size() > 0) {
      const size_t n = size_t(81) * ex_nnz_;
      hipLaunchKernelGGL(rba::k_narrow_matrix, dim3(unsigned(std::min<size_t>((n + 1023) / 1024, 4096))), dim3(256), 0, stream_,
                         static_cast<const double*>(d_ex_vals_.get()), d_ex_vals32_.get(), n);
    }
    if (measure) {
      HIP_CHECK(hipEventRecord(ev_asm1_, stream_));
      asm_pending_ = true;
    }
    ex_valid_ = true;
  }

  // the ranks' partial sums of the assembled matrix: all of it on every rank (replicated products, persistent kernel), or
  // - products split over the ranks - every rank's own range of block slots only (reduce_ranges)
  void reduce_assembled_matrix() {
    // (not with the power-series preconditioner: its terms multiply the WHOLE matrix on every rank)
    if (split_ && !series_fused() && split_slot_bounds_.size() == size_t(nranks_) + 1 && env_.verify_assembled == 0)
      reduce_ranges(d_ex_vals_.get(), split_slot_bounds_);
    else
      all_reduce(d_ex_vals_.get(), size_t(81) * ex_nnz_);
  }

  // The values of the assembled matrix, always DOUBLE. Double solver: off-diagonal blocks from the records of damped
  // top rows, diagonal blocks = the SCHUR_JACOBI blocks of stage 2 (all-reduced there already). Float solver: every
  // block is re-derived in double from the float factors (kernels_a64.hpp) - a float matrix S + E, |E| ~ eps |S|,
  // costs the PCG its accuracy along near-null directions (eps kappa instead of the eps sqrt(kappa) of the square-root
  // product) however its entries are computed.
  void init_a64() {
    if constexpr (kA64) {
      a64_.n_cams = n_cams_;
      a64_.n_lms = n_lms_;
      a64_.lm_k = prm_.lm_k;
      a64_.lm_obs = prm_.lm_obs;
      a64_.obs_cam = prm_.obs_cam;
      a64_.obs_lm = prm_.obs_lm;
      a64_.cam_obs_off = prm_.cam_obs_off;
      a64_.cam_obs = prm_.cam_obs;
      a64_.JpS = prm_.JpS;
      a64_.JpT = prm_.JpT;
      a64_.Vh = prm_.Vh;
      a64_.tauH = prm_.tauH;
      a64_.R0 = prm_.R0;
      a64_.pose_scaling = prm_.pose_scaling;
      a64_.LQ = d_a64_lq_.get();
      a64_.A = d_a64_A_.get();
      a64_.rec = d_a64_rec_.get();
    }
  }
  // JACOBI / power-series preconditioners of a float solver: Hpp^-1 from blocks summed AND factored in double
  // (k_a64_diag<true>, k_invert_blocks<float, double>; a float Cholesky of the float-accumulated Hpp + lambda I met
  // non-positive pivots on final-13682). The scaled Gram blocks D Hpp D belong to the linearisation point.
  void invert_preconditioner_blocks(S lambda) {
    if constexpr (kA64) {
      if (prm_.jacobi && !sc_) {
        if (!gram64_valid_) {
          init_a64();
          if (d_gram64_.size() == 0) d_gram64_.alloc(size_t(81) * n_cams_);
          hipLaunchKernelGGL(rba::k_a64_diag<true>, dim3(rba::xcd_swizzled_grid(n_cams_)), dim3(256), 0, stream_, a64_,
                             static_cast<const int*>(nullptr), d_gram64_.get());
          all_reduce(d_gram64_.get(), size_t(81) * n_cams_);
          gram64_valid_ = true;
        }
        {  // (evaluated once, ahead of the launch macro)
          unsigned long long* const stamp_ptr_ = stamp();
          hipLaunchKernelGGL((rba::k_invert_blocks<S, double>), dim3((n_cams_ + 63) / 64), dim3(64), 0, stream_,
                             d_gram64_.get(), d_inv_.get(), n_cams_, d_fail_.get(), double(lambda), stamp_ptr_);
        }
        return;
      }
    }
    {  // (evaluated once, ahead of the launch macro)
      unsigned long long* const stamp_ptr_ = stamp();
      hipLaunchKernelGGL((rba::k_invert_blocks<S>), dim3((n_cams_ + 63) / 64), dim3(64), 0, stream_, prm_.blocks,
                         d_inv_.get(), n_cams_, d_fail_.get(), S(0), stamp_ptr_);
    }
  }
  void assemble_values() {
    if constexpr (kA64) {
      init_a64();
      if (!a64_lm_valid_) {  // per linearisation point: tau and the reflector cross products in double
        const int short_end = n_tiles_ > 0 ? imp_end_[4] : 0;  // k <= 32: the wave tiles; longer tracks: a wavefront each
        if (n_tiles_ > 0)
          hipLaunchKernelGGL(rba::k_a64_landmark, dim3((n_tiles_ + 3) / 4), dim3(256), 0, stream_, a64_, prm_.OT,
                             implicit_tiles());
        if (n_lms_ > short_end)
          hipLaunchKernelGGL(rba::k_a64_landmark_wave, dim3((n_lms_ - short_end + 3) / 4), dim3(256), 0, stream_, a64_,
                             short_end, n_lms_);
        a64_lm_valid_ = true;
      }
      hipLaunchKernelGGL(rba::k_a64_obs, dim3(unsigned((n_obs_ + rba::kA64Threads - 1) / rba::kA64Threads)),
                         dim3(rba::kA64Threads), 0, stream_, a64_, int64_t(n_obs_), double(pose_damping_));
      hipLaunchKernelGGL(rba::k_a64_diag<false>, dim3(rba::xcd_swizzled_grid(n_cams_)), dim3(256), 0, stream_, a64_,
                         d_ex_diag_.get(), d_ex_vals_.get());
      if (ex_n_upper_ > 0)
        hipLaunchKernelGGL(rba::k_a64_offdiag, dim3(rba::xcd_swizzled_grid(ex_n_upper_)), dim3(256), 0, stream_, a64_,
                           d_ex_vals_.get(), d_ex_upper_.get(), d_ex_mirror_.get(), d_ex_pair_ptr_.get(),
                           d_ex_pair_oi_.get(), d_ex_pair_oj_.get(), ex_n_upper_);
      reduce_assembled_matrix();  // (diagonal blocks included: local sums so far)
    } else {
      ensure_topd();  // the off-diagonal blocks are built from the 27-scalar rows
      if (ex_n_upper_ > 0) launch_offdiag(prm_.topd, d_ex_vals_.get());
      reduce_assembled_matrix();
      // (the diagonal blocks were all-reduced by stage 2 already)
      if (prm_.want_sdiag)
        hipLaunchKernelGGL((rba::k_ex_copy_diag<S>), dim3((81 * n_cams_ + 255) / 256), dim3(256), 0, stream_,
                           prm_.sdiag, d_ex_diag_.get(), d_ex_vals_.get(), n_cams_);
      else
        hipLaunchKernelGGL((rba::k_ex_set_diag<S>), dim3((81 * n_cams_ + 255) / 256), dim3(256), 0, stream_,
                           prm_.blocks, d_ex_diag_.get(), d_ex_vals_.get(), pose_damping_, n_cams_);
    }
  }

  // the block-CSR matrix the PCG runs on: the explicit Schur complement of the SC backend (solver scalar) or the
  // assembled reduced matrix of the square-root solver (double)
  template <class F>
  void with_matrix(F&& f) {
    if (sc_)
      f(static_cast<const int*>(scp_.cols), static_cast<const S*>(scp_.vals), std::false_type{});
    else  // (half storage: kernels_pcg.hpp)
      f(static_cast<const int*>(d_ex_cols_.get()), static_cast<const double*>(d_ex_vals_.get()), std::true_type{});
  }
  // where the pieces of the last product lie (first items, further items of long rows, half storage: transposed parts)
  rba::QPieces<S> q_pieces() const {
    rba::QPieces<S> qp{};
    qp.qmain = d_qmain_.get();
    qp.qextra = d_qpart_.get();
    qp.extra_ptr = d_item_ptr_.get();
    if (!sc_) {
      qp.tpart = d_tpart_.get();
      qp.low_ptr = d_low_ptr_.get();
    }
    return qp;
  }
  // workgroups of the cost evaluation that are resident at once (occupancy of the kernel x compute units)
  int compute_error_blocks() {
    if (ce_blocks_ == 0) {
      int per_cu = 0;
      const hipError_t e = mixed_ ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&rba::k_compute_error<double>), 256, 0)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&rba::k_compute_error<S>), 256, 0);
      ce_blocks_ = (e == hipSuccess && per_cu > 0) ? std::min(kReduceBlocks, per_cu * std::max(1, n_cus_)) : kReduceBlocks;
    }
    return ce_blocks_;
  }
  // wavefronts of the streaming SpMV: as many as are resident at once (more than 256 registers each: one per SIMD)
  int spmv_stream_waves() const {
    if (env_.spmv_stream_waves_per_cu < 0) return 2;  // (tests: two wavefronts walk the whole matrix)
    const int per_cu = env_.spmv_stream_waves_per_cu > 0 ? env_.spmv_stream_waves_per_cu : 4;
    return std::max(1, n_cus_) * per_cu;
  }
  // the row-staged SpMV of kernels_pcg.hpp on the PCG's matrix, MODE 0 / 1 / 2
  // `series_term`: a product INSIDE the power-series preconditioner z = sum_i (Hpp^-1 E0)^i Hpp^-1 r
  // (preconditioner.hpp:180-245). The preconditioner is any fixed symmetric positive definite approximation of the
  // inverse - the float32 reference applies the series in float32 throughout - so those products (ten of the eleven
  // per PCG iteration at power_order 10) stream a FLOAT copy of the assembled matrix, half the bytes; the operator
  // product, the residual refresh and p.q stay on the double matrix (series_f32()).
  template <int MODE>
  void launch_spmv(const S* z, S* p0, S* p1, const S* xvec, const double* part_rho, const double* part_q,
                   double* part_pq, double q_tol, int min_it, int max_it, int period, int* progress,
                   bool series_term = false) {
    auto on_matrix = [&](auto&& f) {
      if constexpr (sizeof(S) == 4) {
        if (series_term && !sc_ && d_ex_vals32_.size() > 0) {
          f(static_cast<const int*>(d_ex_cols_.get()), static_cast<const float*>(d_ex_vals32_.get()), std::true_type{});
          return;
        }
      }
      with_matrix(f);
    };
    on_matrix([&](const int* cols, auto* vals, auto half) {
      using MT = std::remove_cv_t<std::remove_pointer_t<decltype(vals)>>;
      constexpr bool H = decltype(half)::value;
      // (MODE 2 with the products split over the ranks: this rank's range of work items)
      const bool part = MODE == 2 && split_ && split_partial_;
      const int i0 = part ? split_item0_ : 0, ni = part ? split_item1_ - split_item0_ : n_items_;
      if (ni <= 0) return;
      // a matrix of many more items than wavefronts fit the part (nearly dense co-visibility, final-13682): persistent
      // wavefronts that stream their items through a software pipeline (k_pcgs_spmv_stream; bit-identical products)
      bool streamed = false;
      if constexpr (H) {
        const int waves = spmv_stream_waves();
        if (env_.spmv_stream != 0 && (env_.spmv_stream == 2 || ni >= 4 * waves)) {
          streamed = true;
          // (one chunk in flight per wavefront and two wavefronts per SIMD - seven per compute unit, their LDS slots -
          //  against two chunks in flight and one wavefront per SIMD: 24.1 against 26.6 us per float product on
          //  final-13682 (profiles/r6x_*), 92 against 92-96 us per double product on venice-1778+tail (profiles/r6aa_*))
          auto kernel = &rba::k_pcgs_spmv_stream<S, MODE, MT>;
          int launch_waves = waves;
          if (env_.spmv_stream_buffers == 1) {
            kernel = &rba::k_pcgs_spmv_stream1<S, MODE, MT>;
            if (env_.spmv_stream_waves_per_cu == 0) launch_waves = std::max(1, n_cus_) * 7;
          }
          hipLaunchKernelGGL(kernel, dim3(std::min(ni, launch_waves)), dim3(64),
                             size_t(rba::kSpmvPass) * 1024, stream_, cols, vals, d_items_.get() + i0, ni, z, p0, p1, xvec,
                             d_qmain_.get(), d_qpart_.get(), d_tpart_.get(), static_cast<const int*>(d_tdst_.get()),
                             d_cg_.get(), part_rho, part_q, part_pq, q_tol, min_it, max_it, period, progress);
        }
      }
      if (!streamed)
        hipLaunchKernelGGL((rba::k_pcgs_spmv<S, MODE, MT, H>), dim3(ni), dim3(64), rba::spmv_lds_bytes<MT>(), stream_,
                           cols, vals, d_items_.get() + i0, z, p0, p1, xvec, d_qmain_.get(), d_qpart_.get(),
                           H ? d_tpart_.get() : static_cast<double*>(nullptr),
                           H ? d_tdst_.get() : static_cast<const int*>(nullptr), d_cg_.get(), part_rho, part_q, part_pq, q_tol,
                           min_it, max_it, period, progress);
      if (H && n_heavy_ > 0)
        hipLaunchKernelGGL((rba::k_pcgs_reduce_slots<S>), dim3((n_heavy_ + 3) / 4), dim3(256), 0, stream_, d_heavy_.get(),
                           n_heavy_, d_tpart_.get(), d_qpart_.get(), d_cg_.get(), MODE, period);
    });
  }

  // Block structure of the reduced camera matrix: every ordered pair of cameras that
  // observe a common landmark (what BlockSparseMatrix::add ends up holding,
  // block_sparse_matrix.hpp), as block-CSR.
  // (the union over the ranks' landmark shards when sharded; the pair lists are the LOCAL observation pairs)
  void build_sc_structure() {
    const std::vector<int64_t>& lm_obs = h_lm_obs_;
    const std::vector<int>& s_obs_cam = h_obs_cam_;
    const std::vector<std::vector<int>>& nb = ex_nb_;
    const size_t nc = size_t(n_cams_);
    // full rows: the neighbours and the diagonal block (always present: pose damping), ascending column
    std::vector<int> row_ptr(nc + 1, 0), cols, diag(nc);
    for (size_t c = 0; c < nc; ++c) {
      row_ptr[c] = int(cols.size());
      const auto mid = std::lower_bound(nb[c].begin(), nb[c].end(), int(c));
      cols.insert(cols.end(), nb[c].begin(), mid);
      diag[c] = int(cols.size());
      cols.push_back(int(c));
      cols.insert(cols.end(), mid, nb[c].end());
    }
    const int nnz = int(cols.size());
    row_ptr[nc] = nnz;
    auto slot_of = [&](int c, int d) {
      return int(std::lower_bound(cols.begin() + row_ptr[c], cols.begin() + row_ptr[c + 1], d) - cols.begin());
    };
    sc_nnz_ = nnz;
    build_spmv_items(row_ptr, rba::spmv_chunk_blocks<S>() * rba::kSpmvChunksPerItem, nullptr);
    // (persistent PCG: the Schur complement is stored in full, in the solver's scalar)
    build_pcgp_structure(nb, [&](int c, int k) { return 2 * (k < 0 ? diag[c] : slot_of(int(c), nb[c][k])); });
    // upper blocks (ci <= cj; cameras ascend inside a landmark, so i <= j) and, per upper
    // block, the list of contributing observation pairs (counting sort, landmark order)
    std::vector<int> upper_of(size_t(nnz), -1), upper_slot, mirror_slot;
    for (size_t c = 0; c < nc; ++c)
      for (int t = diag[c]; t < row_ptr[c + 1]; ++t) {
        upper_of[t] = int(upper_slot.size());
        upper_slot.push_back(t);
        mirror_slot.push_back(cols[t] == int(c) ? -1 : slot_of(cols[t], int(c)));
      }
    const int n_upper = int(upper_slot.size());
    sc_n_upper_ = n_upper;
    std::vector<int64_t> pair_ptr(size_t(n_upper) + 1, 0);
    // (for a fixed i the blocks (cam_i, cam_j), j >= i, are found by one forward walk of row cam_i)
    auto for_each_pair = [&](auto&& f) {
      for (int l = 0; l < n_lms_; ++l) {
        const int64_t o0 = lm_obs[l];
        const int k = int(lm_obs[l + 1] - o0);
        for (int i = 0; i < k; ++i) {
          const int ci = s_obs_cam[o0 + i];
          int t = diag[ci];
          for (int j = i; j < k; ++j) {
            const int cj = s_obs_cam[o0 + j];
            while (cols[t] < cj) ++t;
            f(upper_of[t], int(o0 + i), int(o0 + j));
          }
        }
      }
    };
    for_each_pair([&](int u, int, int) { ++pair_ptr[size_t(u) + 1]; });
    for (int t = 0; t < n_upper; ++t) pair_ptr[t + 1] += pair_ptr[t];
    const int64_t n_pairs = pair_ptr[n_upper];
    std::vector<int> pair_oi(n_pairs), pair_oj(n_pairs);
    {
      std::vector<int64_t> fill(pair_ptr.begin(), pair_ptr.end() - 1);
      for_each_pair([&](int u, int oi, int oj) {
        const int64_t d = fill[u]++;
        pair_oi[d] = oi;
        pair_oj[d] = oj;
      });
    }
    d_sc_upper_.alloc(n_upper);
    d_sc_mirror_.alloc(n_upper);
    d_sc_upper_.upload(upper_slot.data(), n_upper, stream_);
    d_sc_mirror_.upload(mirror_slot.data(), n_upper, stream_);
    d_sc_pair_ptr_.alloc(pair_ptr.size());
    d_sc_pair_oi_.alloc(n_pairs);
    d_sc_pair_oj_.alloc(n_pairs);
    d_sc_pair_ptr_.upload(pair_ptr.data(), pair_ptr.size(), stream_);
    d_sc_pair_oi_.upload(pair_oi.data(), n_pairs, stream_);
    d_sc_pair_oj_.upload(pair_oj.data(), n_pairs, stream_);
    sc_assemble_bytes_ = n_pairs * int64_t(8 + 54 * sizeof(S)) + int64_t(81) * nnz * sizeof(S);
    d_sc_rowptr_.alloc(row_ptr.size());
    d_sc_cols_.alloc(cols.size());
    d_sc_diag_.alloc(diag.size());
    d_sc_rowptr_.upload(row_ptr.data(), row_ptr.size(), stream_);
    d_sc_cols_.upload(cols.data(), cols.size(), stream_);
    d_sc_diag_.upload(diag.data(), diag.size(), stream_);
    d_sc_vals_.alloc(size_t(81) * nnz + 4);
    scp_.row_ptr = d_sc_rowptr_.get();
    scp_.cols = d_sc_cols_.get();
    scp_.diag_slot = d_sc_diag_.get();
    scp_.vals = d_sc_vals_.get();
    HIP_CHECK(hipStreamSynchronize(stream_));  // the host vectors above go out of scope
    // algorithmic traffic of one S x: the blocks, their column indices, x and y
    hx_bytes_ = int64_t(sizeof(S)) * 81 * nnz + int64_t(4) * nnz + int64_t(sizeof(S)) * 2 * 9 * n_cams_;
    hx_flops_ = int64_t(162) * nnz;
  }

  ~Solver() override { release_resources(); }

  // streams, events, pinned memory, communicator (device buffers are DevBuf members);
  // safe on a partially constructed object
  void release_resources() {
    (void)hipSetDevice(device_);
    destroy_pcg_graphs();
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (comm_ && g_rccl.CommDestroy && !comm_aborted_.load()) g_rccl.CommDestroy(comm_);  // (an aborted one is gone)
    comm_ = nullptr;
    for (auto& e : hx_events_)
      if (e) (void)hipEventDestroy(e);
    hx_events_.clear();
    for (auto& e : sub_events_)
      if (e) (void)hipEventDestroy(e);
    sub_events_.clear();
    for (auto& e : comm_events_)
      if (e) (void)hipEventDestroy(e);
    comm_events_.clear();
    for (auto& e : timer_pool_)
      if (e) (void)hipEventDestroy(e);
    timer_pool_.clear();
    for (hipEvent_t* e : {&ev_asm0_, &ev_asm1_, &ev_fork_, &ev_join_}) {
      if (*e) (void)hipEventDestroy(*e);
      *e = nullptr;
    }
    if (h_pinned_) (void)hipHostFree(h_pinned_);
    h_pinned_ = nullptr;
    if (h_vec_stage_) (void)hipHostFree(h_vec_stage_);
    h_vec_stage_ = nullptr;
    if (h_progress_) (void)hipHostFree(h_progress_);
    if (h_stamps_) (void)hipHostFree(h_stamps_);
    h_stamps_ = nullptr;
    h_progress_ = nullptr;
    if (side_stream_) {
      (void)hipStreamSynchronize(side_stream_);
      (void)hipStreamDestroy(side_stream_);
    }
    side_stream_ = nullptr;
    if (stream_) (void)hipStreamDestroy(stream_);
    stream_ = nullptr;
  }

  // ---- multi-GPU ------------------------------------------------------------
  void comm_init(int rank, int nranks, const void* uid) override {
    // nranks == 1 is allowed on purpose: a one-rank communicator exercises the whole
    // RCCL call path (dlopen, unique id, every all-reduce site) on a single-GPU box
    if (nranks < 1) return;
    if (!g_rccl.load()) throw HipError{"cannot load librccl.so", RBA_ERR_COMM};
    Rccl::UniqueId id;
    std::memcpy(&id, uid, sizeof(id));
    HIP_CHECK(hipSetDevice(device_));
    const int rc = g_rccl.CommInitRank(&comm_, nranks, id, rank);
    if (rc != 0) throw HipError{"ncclCommInitRank failed: " + std::to_string(rc), RBA_ERR_COMM};
    rank_ = rank;
    nranks_ = nranks;
    comm_events_.assign(2 * kCommEvents, nullptr);  // (not lazily inside a timed stage)
    for (auto& e : comm_events_) HIP_CHECK(hipEventCreate(&e));
    union_structure_over_ranks();
    agree_on_persistent_kernel();
    decide_product_split();
  }

  // Whether PCG solves on the assembled matrix run as the persistent kernel is decided per rank from ITS device (the
  // matrix must fit one workgroup per compute unit: n_cus_) - ranks on devices with different CU counts or masks would
  // disagree, and the collective fallback of such a solve (all_reduce of the "gave up" flags) is entered only by ranks
  // that run the kernel: a mismatched collective. One min-reduce makes the decision common.
  void agree_on_persistent_kernel() {
    if (nranks_ <= 1) return;
    int not_ready = pg_ready_ ? 0 : 1;
    d_scratch_int_.upload(&not_ready, 1, stream_);
    all_reduce(d_scratch_int_.get(), 1, kNcclMax);
    d_scratch_int_.download(&not_ready, 1, stream_);
    sync();
    if (not_ready && pg_ready_) {
      if (env_.verbose)
        std::fprintf(stderr, "[rootba_hip] rank %d: the assembled matrix does not fit another rank's register files: all ranks "
                             "use the two-launch PCG\n", rank_);
      pg_ready_ = false;
    }
  }

  // Whether the reduced matrix is assembled at all is decided per rank from ITS shard (pair-list budget): ranks that
  // disagreed would enter different collectives (171 vs 90 n_c in stage 2, the matrix itself, the structure union)
  // and deadlock. The decision is made collective here: if any rank cannot hold its pair lists, every rank drops the
  // assembled operator and stays matrix-free.
  void agree_on_explicit_matrix() {
    if (nranks_ <= 1) return;
    int veto = ex_ready_ ? 0 : 1;
    d_scratch_int_.upload(&veto, 1, stream_);
    all_reduce(d_scratch_int_.get(), 1, kNcclMax);
    d_scratch_int_.download(&veto, 1, stream_);
    sync();
    if (veto && ex_ready_) {
      if (env_.verbose)
        std::fprintf(stderr, "[rootba_hip] rank %d: another rank's pair lists exceed the budget: all ranks stay matrix-free\n",
                     rank_);
      destroy_pcg_graphs();
      ex_ready_ = false;
      ex_valid_ = false;
      prm_.want_sdiag = 0;
      ex_nb_.clear();
      ex_nb_.shrink_to_fit();
    }
  }

  // every rank must hold the SAME block structure for the explicit reduced matrix: union of
  // the co-observation marks of all landmark shards
  void union_structure_over_ranks() {
    if (nranks_ <= 1) return;
    if (!sc_) {
      agree_on_explicit_matrix();
      if (!ex_ready_) return;
    }
    // an all-gather of the ranks' upper pairs (c < d) written as two sum all-reduces (the collective the callback
    // transport has): the pair counts, then one buffer in which every rank fills its own segment
    const size_t nc = size_t(n_cams_);
    std::vector<int> counts(size_t(nranks_), 0);
    for (size_t c = 0; c < nc; ++c)
      counts[size_t(rank_)] += int(ex_nb_[c].end() - std::upper_bound(ex_nb_[c].begin(), ex_nb_[c].end(), int(c)));
    DevBuf<int> d;
    d.alloc(std::max<size_t>(counts.size(), 1));
    d.upload(counts.data(), counts.size(), stream_);
    all_reduce(d.get(), counts.size());
    d.download(counts.data(), counts.size(), stream_);
    sync();
    size_t total = 0, mine = 0;
    for (int r = 0; r < nranks_; ++r) {
      if (r == rank_) mine = total;
      total += size_t(counts[size_t(r)]);
    }
    std::vector<int> pairs(2 * total, 0);
    {
      size_t w = 2 * mine;
      for (size_t c = 0; c < nc; ++c)
        for (int e : ex_nb_[c])
          if (e > int(c)) {
            pairs[w++] = int(c);
            pairs[w++] = e;
          }
    }
    d.alloc(std::max<size_t>(pairs.size(), 1));
    d.upload(pairs.data(), pairs.size(), stream_);
    all_reduce(d.get(), pairs.size());
    d.download(pairs.data(), pairs.size(), stream_);
    sync();
    for (auto& list : ex_nb_) list.clear();
    for (size_t t = 0; t < total; ++t) {
      ex_nb_[size_t(pairs[2 * t])].push_back(pairs[2 * t + 1]);
      ex_nb_[size_t(pairs[2 * t + 1])].push_back(pairs[2 * t]);
    }
    for (auto& list : ex_nb_) {
      std::sort(list.begin(), list.end());
      list.erase(std::unique(list.begin(), list.end()), list.end());
    }
    if (sc_)
      build_sc_structure();
    else
      build_explicit_structure();
  }

  // Products on the assembled matrix SPLIT over the ranks (more than one rank, large matrices). Replicated, every rank
  // streams the whole matrix per PCG iteration and needs no collective (venice-1778: 35 MB, ~10 us - an all-reduce of
  // the 64 KB product vector over xGMI would cost more than it saves); at final-13682 size the matrix is ~1 GB, a product
  // a quarter of a millisecond, and N ranks that each multiply 1/N of the block work items and all-reduce the 0.5 MB
  // vector of partial products are faster. The iteration then runs in the protocol of the matrix-free products
  // (direction kernel, product, all-reduce, update kernel: one collective per iteration, like them); everything but
  // the product stays replicated and bit-identical on all ranks. The decision is a function of the (united) structure
  // and the rank count only - identical on every rank.
  void decide_product_split() {
    split_ = false;
    if (nranks_ <= 1 || !ex_ready_ || sc_) return;
    const double t_product = double(ex_nnz_) * 81 * sizeof(double) / 3.5e12;           // replicated, ~3.5 TB/s measured
    const double t_collective = 30e-6 + double(nvec_) * sizeof(S) * 2 / 50e9;           // latency + ring volume
    split_ = env_.pcg_split >= 0 ? env_.pcg_split != 0 : t_product * (1.0 - 1.0 / nranks_) > t_collective + 10e-6;
    if (!split_) return;
    // contiguous ranges of work items with equal block counts
    std::vector<rba::SpmvItem> items{size_t(n_items_)};
    d_items_.download(items.data(), items.size(), stream_);
    sync();
    std::vector<int64_t> before(size_t(n_items_) + 1, 0);  // blocks in the work items [0, i)
    for (int i = 0; i < n_items_; ++i) before[i + 1] = before[i] + (items[i].slot1 - items[i].slot0);
    auto boundary = [&](int r) {  // first work item of rank r
      const int64_t target = before[n_items_] * r / nranks_;
      return int(std::lower_bound(before.begin(), before.end(), target) - before.begin());
    };
    split_item0_ = rank_ == 0 ? 0 : boundary(rank_);
    split_item1_ = rank_ == nranks_ - 1 ? n_items_ : boundary(rank_ + 1);
    // the ranks' ranges of block slots (work items are in slot order), in matrix scalars: what reduce_ranges sums
    split_slot_bounds_.assign(size_t(nranks_) + 1, 0);
    for (int r = 1; r < nranks_; ++r) {
      const int i = boundary(r);
      split_slot_bounds_[size_t(r)] = size_t(81) * size_t(i < n_items_ ? items[size_t(i)].slot0 : ex_nnz_);
    }
    split_slot_bounds_[size_t(nranks_)] = size_t(81) * size_t(ex_nnz_);
    if (env_.verbose)
      std::fprintf(stderr, "[rootba_hip] rank %d: products on the assembled matrix split over %d ranks: work items %d..%d of %d\n",
                   rank_, nranks_, split_item0_, split_item1_, n_items_);
  }

  void comm_init_callback(int rank, int nranks, rba_allreduce_fn fn, void* ctx) override {
    rank_ = rank;
    nranks_ = nranks;
    cb_fn_ = fn;
    cb_ctx_ = ctx;
    union_structure_over_ranks();
    agree_on_persistent_kernel();
    decide_product_split();
  }

  void comm_info(int* rank, int* nranks, int* transport) override {
    *rank = rank_;
    *transport = comm_ ? 1 : (cb_fn_ ? 2 : 0);
    *nranks = nranks_;
    if (comm_ && g_rccl.CommCount) {
      int n = 0;
      if (g_rccl.CommCount(comm_, &n) == 0) *nranks = n;
    }
  }
  void comm_stats(int64_t* calls, int64_t* bytes, double* seconds) override {
    use_device();
    if (comm_ev_tail_ != comm_ev_head_) sync();
    drain_comm_events();
    *calls = comm_calls_;
    *bytes = comm_bytes_;
    *seconds = comm_seconds_;
    if (comm_untimed_ > 0 && comm_timed_ > 0)  // collectives that found the ring full: priced at the timed ones' mean
      *seconds += comm_seconds_ / double(comm_timed_) * double(comm_untimed_);
  }
  // Elapsed time of the collectives: a RING of event pairs on the solver stream. Completed pairs are drained with
  // hipEventQuery - never a blocking wait inside a solve (round 2 synchronised the stream every 64 collectives, in the
  // middle of the matrix-free PCG it was measuring); a collective that finds the ring full goes untimed.
  void drain_comm_events() {
    while (comm_ev_tail_ != comm_ev_head_) {
      const int i = comm_ev_tail_ % kCommEvents;
      if (hipEventQuery(comm_events_[2 * i + 1]) != hipSuccess) break;
      float ms = 0;
      if (hipEventElapsedTime(&ms, comm_events_[2 * i], comm_events_[2 * i + 1]) == hipSuccess) {
        comm_seconds_ += double(ms) * 1e-3;
        ++comm_timed_;
      }
      ++comm_ev_tail_;
    }
  }

  void comm_abort() override {
    if (comm_aborted_.exchange(true)) return;
    if (comm_ && g_rccl.CommAbort) (void)g_rccl.CommAbort(comm_);  // (in-flight collectives of this rank return)
  }

  template <class T>
  void all_reduce(T* buf, size_t count, int op = kNcclSum) {
    if (!comm_ && !cb_fn_) return;
    if (comm_aborted_.load()) throw HipError{"the communicator was aborted: another rank of this handle failed", RBA_ERR_COMM};
    ++comm_calls_;
    comm_bytes_ += int64_t(count * sizeof(T));
    if (cb_fn_) {
      const double t0 = wall_seconds();
      all_reduce_callback(buf, count, op);
      comm_seconds_ += wall_seconds() - t0;
      return;
    }
    drain_comm_events();
    const bool timed = comm_ev_head_ - comm_ev_tail_ < kCommEvents;
    const int slot = comm_ev_head_ % kCommEvents;
    if (timed) HIP_CHECK(hipEventRecord(comm_events_[2 * slot], stream_));
    const int dt = std::is_same<T, float>::value    ? kNcclFloat32
                   : std::is_same<T, double>::value ? kNcclFloat64
                                                    : kNcclInt32;
    const int rc = g_rccl.AllReduce(buf, buf, count, dt, op, comm_, stream_);
    if (rc != 0) throw HipError{"ncclAllReduce failed: " + std::to_string(rc), RBA_ERR_COMM};
    if (timed) {
      HIP_CHECK(hipEventRecord(comm_events_[2 * slot + 1], stream_));
      ++comm_ev_head_;
    } else {
      ++comm_untimed_;
    }
  }

  template <class T>
  void all_reduce_callback(T* buf, size_t count, int op) {
    {
      // caller-provided collective on a host staging buffer (MPI, gloo, ...)
      cb_stage_.resize(count * sizeof(T));
      HIP_CHECK(hipMemcpyAsync(cb_stage_.data(), buf, count * sizeof(T), hipMemcpyDeviceToHost, stream_));
      sync();
      const int dt = std::is_same<T, float>::value ? 0 : std::is_same<T, double>::value ? 1 : 2;
      const int rc = cb_fn_(cb_ctx_, cb_stage_.data(), int64_t(count), dt, op == kNcclMax ? 1 : 0);
      if (rc != 0) throw HipError{"all-reduce callback failed: " + std::to_string(rc), RBA_ERR_COMM};
      HIP_CHECK(hipMemcpyAsync(buf, cb_stage_.data(), count * sizeof(T), hipMemcpyHostToDevice, stream_));
      sync();
    }
  }

  // Sum over the ranks where every rank needs only ITS range of the result: `bounds` (nranks + 1 element offsets into
  // buf, identical on all ranks) - rank r ends up with the sums of [bounds[r], bounds[r + 1]), the rest of buf keeps this
  // rank's own partial sums. The assembled matrix of products SPLIT over the ranks (decide_product_split): every rank
  // multiplies its range of work items only, so an all-reduce of the whole matrix - reduce-scatter + all-gather - moved
  // twice the bytes it had to (final-13682: 370 MB of half storage per assembly and rank). One ncclReduce per root inside
  // a group (the ranges are balanced by blocks, not equal in size, which ncclReduceScatter would need). The callback
  // transport offers an all-reduce only and keeps it.
  void reduce_ranges(double* buf, const std::vector<size_t>& bounds) {
    if (!comm_ && !cb_fn_) return;
    const size_t total = bounds.back();
    if (cb_fn_ || !g_rccl.Reduce || !g_rccl.GroupStart || !g_rccl.GroupEnd) {
      all_reduce(buf, total);
      return;
    }
    if (comm_aborted_.load()) throw HipError{"the communicator was aborted: another rank of this handle failed", RBA_ERR_COMM};
    ++comm_calls_;
    comm_bytes_ += int64_t((bounds[size_t(rank_) + 1] - bounds[size_t(rank_)]) * sizeof(double));
    int rc = g_rccl.GroupStart();
    for (int r = 0; r < nranks_ && rc == 0; ++r) {
      const size_t n = bounds[size_t(r) + 1] - bounds[size_t(r)];
      if (n > 0) rc = g_rccl.Reduce(buf + bounds[size_t(r)], buf + bounds[size_t(r)], n, kNcclFloat64, kNcclSum, r, comm_, stream_);
    }
    const int rc_end = g_rccl.GroupEnd();
    if (rc != 0 || rc_end != 0) throw HipError{"ncclReduce (ranges of the assembled matrix) failed: " + std::to_string(rc ? rc : rc_end), RBA_ERR_COMM};
  }

  // ---- state ------------------------------------------------------------------
  void set_state(const void* cams, const void* lms) override {
    lm_.ri_is_current = false;  // (the error cached by the LM loop belongs to another state)
    use_device();
    if (mixed_) {
      // RBA_MIXED: the state crosses the boundary in double; the float state is its rounding
      const double* l = static_cast<const double*>(lms);
      std::vector<double> sorted(3 * size_t(n_lms_));
      for (int s = 0; s < n_lms_; ++s)
        for (int c = 0; c < 3; ++c) sorted[3 * size_t(s) + c] = l[3 * size_t(perm_[s]) + c];
      d_cams64_.upload(static_cast<const double*>(cams), 10 * size_t(n_cams_), stream_);
      d_lms64_.upload(sorted.data(), sorted.size(), stream_);
      round_masters();
      sync();
      return;
    }
    const S* l = static_cast<const S*>(lms);
    std::vector<S> sorted(3 * size_t(n_lms_));
    for (int s = 0; s < n_lms_; ++s)
      for (int c = 0; c < 3; ++c) sorted[3 * size_t(s) + c] = l[3 * size_t(perm_[s]) + c];
    d_cams_.upload(static_cast<const S*>(cams), 10 * size_t(n_cams_), stream_);
    d_lms_.upload(sorted.data(), sorted.size(), stream_);
    sync();
  }
  // (either pointer may be null: cameras only / landmarks only)
  void get_state(void* cams, void* lms) override {
    use_device();
    if (mixed_) {
      std::vector<double> sorted(lms ? 3 * size_t(n_lms_) : 0);
      if (cams) d_cams64_.download(static_cast<double*>(cams), 10 * size_t(n_cams_), stream_);
      if (lms) d_lms64_.download(sorted.data(), sorted.size(), stream_);
      sync();
      double* l = static_cast<double*>(lms);
      if (lms)
        for (int s = 0; s < n_lms_; ++s)
          for (int c = 0; c < 3; ++c) l[3 * size_t(perm_[s]) + c] = sorted[3 * size_t(s) + c];
      return;
    }
    std::vector<S> sorted(lms ? 3 * size_t(n_lms_) : 0);
    if (cams) d_cams_.download(static_cast<S*>(cams), 10 * size_t(n_cams_), stream_);
    if (lms) d_lms_.download(sorted.data(), sorted.size(), stream_);
    sync();
    S* l = static_cast<S*>(lms);
    if (lms)
      for (int s = 0; s < n_lms_; ++s)
        for (int c = 0; c < 3; ++c) l[3 * size_t(perm_[s]) + c] = sorted[3 * size_t(s) + c];
  }
  void backup() override {
    use_device();
    HIP_CHECK(hipMemcpyAsync(d_cams_bak_.get(), d_cams_.get(), d_cams_.size() * sizeof(S),
                             hipMemcpyDeviceToDevice, stream_));
    HIP_CHECK(hipMemcpyAsync(d_lms_bak_.get(), d_lms_.get(), d_lms_.size() * sizeof(S),
                             hipMemcpyDeviceToDevice, stream_));
    if (mixed_) {
      HIP_CHECK(hipMemcpyAsync(d_cams64_bak_.get(), d_cams64_.get(), d_cams64_.size() * sizeof(double),
                               hipMemcpyDeviceToDevice, stream_));
      HIP_CHECK(hipMemcpyAsync(d_lms64_bak_.get(), d_lms64_.get(), d_lms64_.size() * sizeof(double),
                               hipMemcpyDeviceToDevice, stream_));
    }
  }
  // float state = rounding of the double masters (RBA_MIXED)
  void round_masters() {
    const int64_t nc = 10 * int64_t(n_cams_), nl = 3 * int64_t(n_lms_);
    hipLaunchKernelGGL(rba::k_mixed_round, dim3(unsigned((nc + 255) / 256)), dim3(256), 0, stream_, d_cams64_.get(),
                       reinterpret_cast<float*>(d_cams_.get()), nc);
    hipLaunchKernelGGL(rba::k_mixed_round, dim3(unsigned((nl + 255) / 256)), dim3(256), 0, stream_, d_lms64_.get(),
                       reinterpret_cast<float*>(d_lms_.get()), nl);
  }
  void restore() override {
    if (!in_lm_step_) lm_.ri_is_current = false;  // (see apply())
    use_device();
    HIP_CHECK(hipMemcpyAsync(d_cams_.get(), d_cams_bak_.get(), d_cams_.size() * sizeof(S),
                             hipMemcpyDeviceToDevice, stream_));
    HIP_CHECK(hipMemcpyAsync(d_lms_.get(), d_lms_bak_.get(), d_lms_.size() * sizeof(S),
                             hipMemcpyDeviceToDevice, stream_));
    if (mixed_) {
      HIP_CHECK(hipMemcpyAsync(d_cams64_.get(), d_cams64_bak_.get(), d_cams64_.size() * sizeof(double),
                               hipMemcpyDeviceToDevice, stream_));
      HIP_CHECK(hipMemcpyAsync(d_lms64_.get(), d_lms64_bak_.get(), d_lms64_.size() * sizeof(double),
                               hipMemcpyDeviceToDevice, stream_));
    }
  }

  // ---- compute_error ----------------------------------------------------------
  void compute_error(rba_residual_info* out) override {
    use_device();
    double* h = pinned_doubles(kPinCe0);
    compute_error_enqueue(h);
    if (lm_async_) sync();  // (a direct call from inside the LM loop: not used by lm_step itself)
    compute_error_parse(h, out);
  }
  // kernels + the copy of the eight sums into pinned host memory `h`; valid after the next synchronisation
  // `side`: on the side stream, concurrently with what follows on the solver stream (rba_lm_step: the cost of the current
  // state - which the reference re-evaluates at every outer iteration - needs nothing but the state, like the
  // linearisation that is queued right behind it: a 0.3-of-roofline gather pass beside a streaming one); joined before the
  // state is touched again (join_side) and at every synchronisation. One rank only (no collective on the side stream).
  void compute_error_enqueue(double* h, bool side = false) {
    ++pcg_counters_.cost_evaluations;
    side = side && results_go_direct();
    hipStream_t st = side ? side_stream_ : stream_;
    double* part = side ? d_partials_side_.get() : d_partials_.get();
    if (side) {
      HIP_CHECK(hipEventRecord(ev_fork_, stream_));
      HIP_CHECK(hipStreamWaitEvent(side_stream_, ev_fork_, 0));
    }
    time_begin(st);
    // (device stamps: the evaluation kernel stamps its start - the pending boundary of the solver stream, or the side
    //  stream's own - and the reduction kernel, a single workgroup, the stage's end)
    unsigned long long* s_begin = side ? stamp_side_ : stamp();
    stamp_side_ = nullptr;
    const int end_slot = stamps_on() ? stamp_slot() : -1;
    unsigned long long* s_end = end_slot >= 0 ? h_stamps_ + end_slot : nullptr;
    // as many workgroups as are RESIDENT at once (the kernel is a chain of dependent gathers per work-item: with 2048
    // workgroups at the 6 wavefronts per SIMD its 74 registers allow, a quarter of them ran as a second round on a third of
    // the part - twenty round trips per work-item's share instead of thirteen)
    const int blocks = int(std::min<int64_t>(compute_error_blocks(), (n_obs_ + 255) / 256));
    // the eight sums: one rank - straight into the pinned host page; more ranks - into `red`, all-reduced below, or, at
    // the end of an iteration of rba_lm_step (lm_merge_end_), into the block that carries l_diff and the failure bits as
    // well (one collective)
    const bool direct = results_go_direct();
    const bool merged = !direct && lm_merge_end_ && !side;
    double* red = merged ? d_endred_.get() : part + size_t(kReduceBlocks) * 8;
    if (mixed_) {
      rba::Params<double> p64 = prm64_;
      p64.stamp = s_begin;
      hipLaunchKernelGGL((rba::k_compute_error<double>), dim3(blocks), dim3(256), 0, st, p64, n_obs_, part);
    } else {
      rba::Params<S> ps = prm_;
      ps.stamp = s_begin;
      hipLaunchKernelGGL((rba::k_compute_error<S>), dim3(blocks), dim3(256), 0, st, ps, n_obs_, part);
    }
    hipLaunchKernelGGL((rba::k_reduce_rows<8>), dim3(1), dim3(256), 0, st, part, int64_t(blocks), red,
                       direct ? h : static_cast<double*>(nullptr), static_cast<int*>(nullptr),
                       static_cast<int*>(nullptr), 0, s_end, static_cast<double*>(nullptr));
    if (merged) {
      all_reduce(d_endred_.get(), size_t(rba::kEndRed));
      HIP_CHECK(hipMemcpyAsync(pinned_doubles(kPinEnd), d_endred_.get(), rba::kEndRed * sizeof(double), hipMemcpyDeviceToHost,
                               stream_));
      lm_end_merged_ = true;
    } else if (!direct) {
      all_reduce(red, 8);
      HIP_CHECK(hipMemcpyAsync(h, red, 8 * sizeof(double), hipMemcpyDeviceToHost, stream_));
    }
    time_end(&timings_.residual_evaluation_time, true, st, end_slot);
    if (side) {
      HIP_CHECK(hipEventRecord(ev_join_, side_stream_));
      side_pending_ = true;
    }
  }
  // the solver stream waits for the side stream's work (a no-op when there is none)
  void join_side() {
    if (!side_pending_) return;
    HIP_CHECK(hipStreamWaitEvent(stream_, ev_join_, 0));
    side_pending_ = false;
  }
  static void compute_error_parse(const double* h, rba_residual_info* out) {
    out->all_num_obs = int(std::llround(h[0]));
    out->all_error = h[1];
    out->all_residual_sum = h[2];
    out->valid_num_obs = int(std::llround(h[3]));
    out->valid_error = h[4];
    out->valid_residual_sum = h[5];
    out->is_numerically_valid = h[6] == 0.0 ? 1 : 0;
  }

  // ---- stage 1 ------------------------------------------------------------------
  bool s1_fused() const { return !sc_ && n_tiles_ > 0 && env_.s1_fused && !sub_timing(); }

  int linearize(void* jp_diag2_out) override {
    if (env_.test_fail_rank >= 0 && nranks_ > 1 && rank_ == env_.test_fail_rank)
      throw HipError{"RBA_TEST_FAIL_RANK: rank " + std::to_string(rank_) + " fails on purpose", RBA_ERR_HIP};
    use_device();
    time_begin();
    sub_begin();
    // (the failure word is clean here: whoever publishes it to the host resets the bits of its phase)
    // (sub-stage timers: the reference's stages one by one - the unfused kernels)
    const bool fuse = s1_fused();
    if (!sc_) {
      // geometry once per observation; Jp_diag2 falls out of the camera-major Gram pass. Wave-tile landmarks
      // (k <= 32): geometry and QR in ONE kernel, an observation per lane (k_s1_fused_obs) - the geometry kernel then
      // only serves the observations of the longer tracks
      // (rounded down to an even observation: the kernel's 16-byte stores stay aligned; the fused kernel, later in the
      //  stream, writes that observation again)
      const int64_t o_begin = fuse ? (n_obs_tiled_ & ~int64_t(1)) : 0;
      if (o_begin < n_obs_) {
        const rba::Params<S> prm_st_ = prm_stamped();
        hipLaunchKernelGGL((rba::k_s1_geometry<S>), dim3(unsigned((n_obs_ - o_begin + 255) / 256)), dim3(256),
                           256 * 26 * sizeof(S), stream_, prm_st_, o_begin, int64_t(n_obs_));
      }
      if (fuse) {
        const rba::FusedObsWaves fw = fused_obs_waves();
        {  // (evaluated once, ahead of the launch macro)
          const rba::Params<S> prm_st_ = prm_stamped();
          hipLaunchKernelGGL((rba::k_s1_fused_obs<S>), dim3((fw.wave_begin[5] + 3) / 4), dim3(256), 0, stream_,
                             prm_st_, implicit_tiles(), fw);
        }
      }
      sub_mark(&sub_.jacobian_evaluation_time);  // linearize_problem()
      // One GPU: the Gram pass (Jp_diag2, pose scaling, B_mid) is folded into the camera-major pass of the first
      // stage 2, which gathers the same Jacobian rows anyway (k_cam_pass*<0>, GRAM). Not when the caller wants
      // Jp_diag2 now, with more than one rank (Jp_diag2 is all-reduced before it is used) or with the unstaged
      // sub-stage timers.
      gram_pending_ = !comm_ && !cb_fn_ && !jp_diag2_out && !sub_timing();
      if (!gram_pending_) launch_cam_gram(prm_);
    } else {
      hipLaunchKernelGGL((rba::k_sc_jp_diag2<S>), dim3(n_cams_), dim3(256), 0, stream_, prm_);
    }
    if (!gram_pending_) all_reduce(d_jp_diag2_.get(), nvec_);
    if (!lm_fuse_) all_reduce(d_fail_.get(), 1, kNcclMax);
    if (!gram_pending_)
      hipLaunchKernelGGL((rba::k_pose_scaling<S>), dim3((nvec_ + 255) / 256), dim3(256), 0, stream_,
                         d_jp_diag2_.get(), d_pose_scaling_.get(), prm_.eps, nvec_);
    if (!sc_) {
      sub_mark(&sub_.scale_landmark_jacobian_time);  // get_Jp_diag2() (+ scale_Jl_cols, done inside the QR kernels)
      if (!gram_pending_)
        hipLaunchKernelGGL((rba::k_scale_gram<S>), dim3((81 * n_cams_ + 255) / 256), dim3(256), 0, stream_, prm_);
      sub_mark(&sub_.stage1_preconditioner_time);  // get_Jp_T_Jp_blockdiag() (JACOBI blocks)
      if (n_tiles_ > 0 && !fuse)
        hipLaunchKernelGGL((rba::k_s1_qr_tile<S>), dim3((n_tiles_ + 3) / 4), dim3(256), 0, stream_, prm_,
                           implicit_tiles());
      if (imp_end_[5] > imp_begin_[5])
        hipLaunchKernelGGL((rba::k_s1_qr_wide<S, 2>), dim3((imp_end_[5] - imp_begin_[5] + 3) / 4), dim3(256), 0,
                           stream_, prm_, imp_begin_[5], imp_end_[5]);
      if (imp_end_[6] > imp_begin_[6])
        hipLaunchKernelGGL((rba::k_s1_qr_wide<S, 4>), dim3((imp_end_[6] - imp_begin_[6] + 3) / 4), dim3(256), 0,
                           stream_, prm_, imp_begin_[6], imp_end_[6]);
      if (n_big_ > 0)
        hipLaunchKernelGGL((rba::k_s1_qr_big<S>), dim3(n_big_), dim3(256), 0, stream_, prm_, big_begin_,
                           d_big_scratch_.get(), d_big_off_.get());
      sub_mark(&sub_.perform_qr_time);  // perform_qr()
    } else {
      // LinearizorSC::linearize (linearizor_sc.cpp:70-99)
      hipLaunchKernelGGL((rba::k_sc_linearize_obs<S>), dim3(unsigned((n_obs_ + 255) / 256)), dim3(256), 0,
                         stream_, scp_);
      hipLaunchKernelGGL((rba::k_sc_landmark_moments<S>), dim3((n_lms_ + 255) / 256), dim3(256), 0, stream_,
                         scp_);
    }
    HIP_CHECK(hipGetLastError());
    int* fail = pinned_int(kPinFailLin);
    if (lm_fuse_) {
      // rba_lm_step: nobody reads the word before the iteration's one synchronisation - it travels with the end of the
      // solve (one rank: k_finish_increment) or with the sums that end the iteration (more ranks: rba::kEndRed; the
      // ranks' bits are summed there instead of max-reduced here)
      lin_flag_deferred_ = true;
    } else if (results_go_direct()) {
      hipLaunchKernelGGL(rba::k_publish_flag, dim3(1), dim3(1), 0, stream_, d_fail_.get(), fail, 1);
    } else {
      HIP_CHECK(hipMemcpyAsync(fail, d_fail_.get(), sizeof(int), hipMemcpyDeviceToHost, stream_));
      hipLaunchKernelGGL(rba::k_publish_flag, dim3(1), dim3(1), 0, stream_, d_fail_.get(), d_scratch_int_.get(), 1);
    }
    if (jp_diag2_out) d_jp_diag2_.download(static_cast<S*>(jp_diag2_out), nvec_, stream_);
    time_end(&timings_.stage1_time);
    sub_collect();
    pose_damping_ = S(0);
    landmark_damping_valid_ = false;
    ex_valid_ = false;
    a64_lm_valid_ = false;
    gram64_valid_ = false;
    if (lm_async_) return RBA_OK;  // lm_step reads the flag at its next synchronisation point (linearize_failed)
    return (*fail & 1) ? RBA_NUMERICAL_FAILURE : RBA_OK;
  }

  // ---- stage 2 ------------------------------------------------------------------
  void run_stage2(S lambda) {
    if (sc_) {
      // set_landmark_damping + get_Hb + block-diagonal copy (linearizor_sc.cpp:101-140)
      hipLaunchKernelGGL((rba::k_sc_landmark_inverse<S>), dim3((n_lms_ + 255) / 256), dim3(256), 0, stream_,
                         scp_, lambda);
      hipLaunchKernelGGL((rba::k_sc_obs_products<S>), dim3(unsigned((n_obs_ + 255) / 256)), dim3(256), 0,
                         stream_, scp_);
      launch_sc_assemble(scp_);
      hipLaunchKernelGGL((rba::k_sc_cam_gradient<S>), dim3(n_cams_), dim3(256), 0, stream_, scp_);
      if (comm_ || cb_fn_) {
        // landmarks sharded: H_pp - sum_l W_l H_ll^-1 W_l^T and the gradient are sums over landmarks - every rank holds
        // the sums of its shard in the united structure; the PCG on the summed matrix then runs replicated on all ranks
        // with no further collective (LinearizationSC sums the same terms, linearization_sc.hpp:232-347)
        all_reduce(d_sc_vals_.get(), size_t(81) * sc_nnz_);
        all_reduce(d_bb_.get(), nvec_);
      }
      hipLaunchKernelGGL((rba::k_sc_damp_and_extract_diag<S>), dim3((81 * n_cams_ + 255) / 256), dim3(256), 0,
                         stream_, scp_, lambda);
      if (opt_.preconditioner_type == 2) {
        // PowerSCPreconditioner on the explicit system (linearizor_sc.cpp:163-170): its blocks are the JACOBI
        // blocks Hpp = Jp^T Jp (+ pose damping), get_jacobi(); the series (Hpp^-1 E0)^i runs through S = Hpp - E0
        // (pcg). The scaled rows of the SC path give D G D directly; Jp_diag2 of this pass goes to a scratch vector.
        rba::Params<S> gp = prm_;
        gp.JpS = scp_.JpS;
        gp.JpT = scp_.JpS + 16 * size_t(n_obs_);
        gp.jp_diag2 = d_tmp_.get();
        launch_cam_gram(gp);
        all_reduce(d_mid_.get(), size_t(81) * n_cams_);
        hipLaunchKernelGGL((rba::k_sc_jacobi_blocks<S>), dim3((81 * n_cams_ + 255) / 256), dim3(256), 0, stream_,
                           prm_.B_mid, lambda, scp_.blocks, n_cams_);
      }
      pose_damping_ = lambda;
      landmark_damping_valid_ = true;
      return;
    }
    sub_begin();
    // landmark side: the six damping rotations per landmark and the stage-2 record of every observation
    // (set_landmark_damping + scale_Jp_cols + the per-column part of the damping, kernels_s1.hpp)
    {  // (evaluated once, ahead of the launch macro)
      const rba::Params<S> prm_st_ = prm_stamped();
      hipLaunchKernelGGL((rba::k_s2_obs<S>), dim3(unsigned((n_obs_ + 255) / 256)), dim3(256), 0, stream_, prm_st_,
                         int64_t(n_obs_), lambda);
    }
    topd_valid_ = false;
    sub_mark(&sub_.scale_pose_jacobian_time);
    launch_cam_stage2(prm_, lambda);
    gram_pending_ = false;
    sub_mark(&sub_.stage2_preconditioner_and_gradient_time);  // get_Q2TJp_T_Q2TJp_blockdiag() + get_Q2TJp_T_Q2Tr()
    if (comm_ || cb_fn_) {
      // every rank added lambda*I and holds only its landmarks' sums: make the
      // diagonal term count once
      all_reduce(d_bb_.get(), size_t(prm_.want_sdiag ? 171 : 90) * n_cams_);
      hipLaunchKernelGGL((rba::k_sub_diag<S>), dim3((nvec_ + 255) / 256), dim3(256), 0, stream_,
                         prm_.blocks, S(lambda) * S(nranks_ - 1), n_cams_);
    }
    pose_damping_ = lambda;
    landmark_damping_valid_ = true;
    ex_valid_ = false;
  }

  int stage2(double lambda, void* b_out, void* blocks_out) override {
    use_device();
    time_begin();
    run_stage2(S(lambda));
    if (b_out) d_bb_.download(static_cast<S*>(b_out), nvec_, stream_);
    if (blocks_out)
      HIP_CHECK(hipMemcpyAsync(blocks_out, prm_.blocks, size_t(81) * n_cams_ * sizeof(S),
                               hipMemcpyDeviceToHost, stream_));
    time_end(&timings_.stage2_time);
    sub_collect();
    return RBA_OK;
  }

  // y += sum_l A_l^T A_l x_l over the local landmarks (no pose damping term)
  void launch_hx(const S* x, S* y, const int* done_flag = nullptr) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // every hx_timing_stride_-th product is bracketed by HIP events (each event is a
    // marker packet on the queue: timing all of them costs ~15 us per PCG iteration)
    // (only the matrix-free product is timed: rba_iter_timings.hx_time is its roofline input)
    if (!ex_active_ && hx_timing_stride_ > 0 &&
        hx_calls_ % hx_timing_stride_ == (hx_timing_stride_ > 1 ? 1 : 0) && hx_event_count_ < kMaxHxEvents) {
      e0 = hx_events_[2 * hx_event_count_];
      e1 = hx_events_[2 * hx_event_count_ + 1];
      hx_event_call_[hx_event_count_] = hx_calls_;
      ++hx_event_count_;
      HIP_CHECK(hipEventRecord(e0, stream_));
    }
    if (ex_active_ || sc_) {
      // y = (sum_l A_l^T A_l) x from the assembled matrix (SC backend: y = S x, the pose damping is part of S);
      // overwrites y. (Callers: the round-1 PCG loop, which still serves the repeat of a solve whose products went
      // back to matrix-free, the power series without an assembled matrix, and the products split over ranks.)
      if (sc_) {
        hipLaunchKernelGGL((rba::k_sc_spmv<S>), dim3(n_cams_), dim3(256), 0, stream_, scp_, x, y, done_flag);
      } else {
        // the row-staged SpMV of the fused PCG in its plain-product mode (double blocks, kernels_pcg.hpp) + its collect.
        // Split over the ranks (decide_product_split): this rank's work items only, into cleared pieces - the caller
        // all-reduces y
        if (split_) {
          d_qmain_.zero(stream_);
          d_qpart_.zero(stream_);
          d_tpart_.zero(stream_);
          split_partial_ = true;
        }
        launch_spmv<2>(nullptr, nullptr, nullptr, x, nullptr, nullptr, nullptr, 0.0, 0, 0, 1, nullptr);
        split_partial_ = false;
        hipLaunchKernelGGL((rba::k_pcgs_collect<S>), dim3((nvec_ + 255) / 256), dim3(256), 0, stream_, y, q_pieces(),
                           nvec_);
      }
      if (e1) HIP_CHECK(hipEventRecord(e1, stream_));
      ++hx_calls_;
      return;
    }
    launch_hx_implicit(x, y, done_flag);
    if (e1) HIP_CHECK(hipEventRecord(e1, stream_));
    ++hx_calls_;
  }

  // SC backend assembly: matrix cores for float, VALU for double
  void launch_sc_assemble(const rba::ScParams<float>& sp) {
    hipLaunchKernelGGL((rba::k_sc_assemble_mfma), dim3(sc_n_upper_), dim3(256), 0, stream_, sp, d_sc_upper_.get(),
                       d_sc_mirror_.get(), d_sc_pair_ptr_.get(), d_sc_pair_oi_.get(), d_sc_pair_oj_.get());
  }
  void launch_sc_assemble(const rba::ScParams<double>& sp) {
    hipLaunchKernelGGL((rba::k_sc_assemble<double>), dim3(sc_n_upper_), dim3(256), 0, stream_, sp,
                       d_sc_upper_.get(), d_sc_mirror_.get(), d_sc_pair_ptr_.get(), d_sc_pair_oi_.get(),
                       d_sc_pair_oj_.get(), sc_n_upper_);
  }

  // off-diagonal blocks of the explicit reduced matrix on the matrix cores of either precision (kernels_sc.hpp)
  void launch_offdiag(const double* topd, double* vals) {
    hipLaunchKernelGGL((rba::k_ex_offdiag_mfma<double>), dim3(rba::xcd_swizzled_grid(ex_n_upper_)), dim3(256), 0, stream_,
                       topd, vals, d_ex_upper_.get(), d_ex_mirror_.get(), d_ex_pair_ptr_.get(), d_ex_pair_oi_.get(),
                       d_ex_pair_oj_.get(), ex_n_upper_);
  }

  // camera-major passes (kernels_cam.hpp) on the matrix cores of either precision.
  // launch_cam_gram: Jp_diag2 + the unscaled Gram blocks on their own (sharded runs, Jp_diag2 requested at once, the SC
  // backend's power-series blocks); launch_cam_stage2: blocks, b (and on one GPU the Gram part of the first stage 2)
  void launch_cam_gram(const rba::Params<S>& prm) {
    hipLaunchKernelGGL((rba::k_cam_pass_mfma<S, 1>), dim3(rba::xcd_swizzled_grid(n_cams_)), dim3(256), 0, stream_, prm,
                       S(0), 0);
  }
  void launch_cam_stage2(const rba::Params<S>& prm, S lambda) {
    hipLaunchKernelGGL((rba::k_cam_pass_mfma<S, 0>), dim3(rba::xcd_swizzled_grid(n_cams_)), dim3(256), 0, stream_, prm,
                       lambda, gram_pending_ ? 1 : 0);
  }
  // the 27 + 9 records of the current damping, for the consumers that read them (assembly of the reduced matrix,
  // matrix-free E0 products): the closed-form column pass on the unscaled rows (kernels_s1.hpp)
  void ensure_topd() {
    if (topd_valid_) return;
    hipLaunchKernelGGL((rba::k_s12_cols<S>),
                       dim3(unsigned((n_obs_ + rba::kS1ColsThreads - 1) / rba::kS1ColsThreads)),
                       dim3(rba::kS1ColsThreads), size_t(rba::kS1ColsThreads) * (18 + rba::kTdLds) * sizeof(S), stream_,
                       prm_, int64_t(n_obs_));
    topd_valid_ = true;
  }
  // the stage-1 Gram pass on its own, for callers that read the pose scaling / Jp_diag2 before any stage 2
  void ensure_gram() {
    if (!gram_pending_) return;
    launch_cam_gram(prm_);
    hipLaunchKernelGGL((rba::k_pose_scaling<S>), dim3((nvec_ + 255) / 256), dim3(256), 0, stream_,
                       d_jp_diag2_.get(), d_pose_scaling_.get(), prm_.eps, nvec_);
    hipLaunchKernelGGL((rba::k_scale_gram<S>), dim3((81 * n_cams_ + 255) / 256), dim3(256), 0, stream_, prm_);
    gram_pending_ = false;
  }
  // x -> D x for the kernels that read the unscaled Jacobian rows
  const S* scaled_operand(const S* x) {
    if (operand_prescaled_) return d_xs_.get();  // the producer of x wrote D x already (k_pcgs_direction)
    unsigned long long* const stamp_ptr_ = stamp();  // (evaluated once, ahead of the launch macro)
    hipLaunchKernelGGL((rba::k_scale_vec<S>), dim3((nvec_ + 255) / 256), dim3(256), 0, stream_, x, prm_.pose_scaling,
                       d_xs_.get(), nvec_, stamp_ptr_);
    return d_xs_.get();
  }

  // y += E0 v over the local landmarks (power-series preconditioner)
  void launch_e0(const S* v, S* y, const int* done_flag) {
    ensure_topd();
    for_each_class([&](auto ch_tag, int begin, int end) {
      constexpr int CH = decltype(ch_tag)::value;
      hipLaunchKernelGGL((rba::k_e0<S, CH>), dim3((end - begin + 3) / 4), dim3(256), 0, stream_,
                         prm_, begin, end, v, y, done_flag, d_e0_w_.get());
    });
    if (n_big_ > 0)
      hipLaunchKernelGGL((rba::k_e0_big<S>), dim3(n_big_), dim3(256), 0, stream_, prm_, big_begin_, v, y,
                         done_flag, d_e0_w_.get());
    if (d_e0_w_.get())  // RBA_DETERMINISTIC=1: the kernels above stored the landmark sums; applied camera-major
      hipLaunchKernelGGL((rba::k_e0_det_gather<S>), dim3(rba::xcd_swizzled_grid(n_cams_)), dim3(256), 0, stream_, prm_,
                         d_e0_w_.get(), y, done_flag);
  }

  rba::ImplicitTiles implicit_tiles() const {
    rba::ImplicitTiles it;
    for (int c = 0; c < 5; ++c) {
      it.tile_begin[c] = imp_tile_begin_[c];
      it.lm_begin[c] = imp_begin_[c];
      it.lm_end[c] = imp_end_[c];
    }
    it.tile_begin[5] = n_tiles_;
    return it;
  }

  rba::FusedObsWaves fused_obs_waves() const {
    rba::FusedObsWaves fw;
    int w = 0;
    for (int c = 0; c < 5; ++c) {
      fw.wave_begin[c] = w;
      w += (imp_tiles_[c] + 1) / 2;
    }
    fw.wave_begin[5] = w;
    return fw;
  }

  // H x from the factors; long tracks use the workgroup-per-landmark kernel.
  // The Jacobian rows are unscaled, so the kernels get D x and multiply what they add to y by D
  // (the LDS kernel where its sums leave the workgroup, the others per scatter-add).
  void launch_hx_implicit(const S* x, S* y, const int* done_flag) {
    const S* xin = scaled_operand(x);
    const S* dout = prm_.pose_scaling;
    S* hx_u = d_hx_u_.get();  // RBA_DETERMINISTIC=1: the landmark-major kernels store row entries, summed camera-major
    if (n_big_ > 0)
      hipLaunchKernelGGL((rba::k_hx_implicit_big<S>), dim3(n_big_), dim3(256), 0, stream_, prm_, big_begin_,
                         d_big_scratch_.get(), d_big_off_.get(), xin, y, dout, done_flag, hx_u);
    const bool use_lds = !hx_u && n_tiles_ > 0 && env_.hx_lds &&
                         (env_.hx_lds == 2 || (n_tiles_ >= 16 * n_cus_ && hx_coverage_ >= 0.9));
    // (landmarks with 32 < k <= 64: inside the persistent kernel when that one runs - one launch less per product)
    const bool wide_inside = use_lds && env_.hx_wide_inside;
    if (imp_end_[6] > imp_begin_[6])
      hipLaunchKernelGGL((rba::k_hx_implicit_wide<S, 4>), dim3((imp_end_[6] - imp_begin_[6] + 3) / 4),
                         dim3(256), 0, stream_, prm_, imp_begin_[6], imp_end_[6], xin, y, dout, done_flag, hx_u);
    if (imp_end_[5] > imp_begin_[5] && !wide_inside)
      hipLaunchKernelGGL((rba::k_hx_implicit_wide<S, 2>), dim3((imp_end_[5] - imp_begin_[5] + 3) / 4),
                         dim3(256), 0, stream_, prm_, imp_begin_[5], imp_end_[5], xin, y, dout, done_flag, hx_u);
    rba::ImplicitTiles it = implicit_tiles();
    if (n_tiles_ > 0 && !use_lds)
      hipLaunchKernelGGL((rba::k_hx_implicit<S>), dim3((n_tiles_ + 3) / 4), dim3(256), 0, stream_, prm_, it, xin, y,
                         dout, done_flag, hx_u);
    if (hx_u)
      hipLaunchKernelGGL((rba::k_hx_det_gather<S>), dim3(rba::xcd_swizzled_grid(n_cams_)), dim3(256), 0, stream_, prm_,
                         hx_u, y, dout, done_flag);
    if (use_lds) {
      // workgroup-private window of y in LDS (double accumulators, ds_add_f64), one persistent workgroup per CU
      const size_t ylds_bytes = size_t(9) * hx_win_ * sizeof(double);
      // double needs 156 VGPRs: 512-thread workgroups (no scratch; venice: 254 us against 428 with 1024 threads capped
      // at 128 VGPRs); float runs 1024 threads at 114 VGPRs
      rba::HxWideRanges wide{{0}, {0}};
      if (wide_inside) wide = rba::HxWideRanges{{imp_begin_[5]}, {imp_end_[5]}};
      launch_hx_lds<(sizeof(S) == 8 ? 512 : 1024)>(it, ylds_bytes, xin, y, dout, done_flag, wide);
    }
  }

  template <int NT>
  void launch_hx_lds(const rba::ImplicitTiles& it, size_t ylds_bytes, const S* xin, S* y, const S* dout,
                     const int* done_flag, const rba::HxWideRanges& wide) {
    // (the wavefronts' buffers of the wide landmarks reuse the window)
    ylds_bytes = std::max(ylds_bytes, size_t(NT / 64) * (rba::kHxWideScalars * sizeof(S) + 128));
    if (!hx_lds_attr_set_) {
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rba::k_hx_implicit_lds<S, NT, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, int(kHxLdsMaxBytes)));
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rba::k_hx_implicit_lds<S, NT, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, int(kHxLdsMaxBytes)));
      hx_lds_attr_set_ = true;
    }
    if (hx_win_ >= n_cams_)  // every camera in the window: the instance without the path of global atomics
      hipLaunchKernelGGL((rba::k_hx_implicit_lds<S, NT, true>), dim3(n_hx_chunks_), dim3(NT), ylds_bytes, stream_, prm_, it,
                         d_hx_chunks_.get(), hx_win_, xin, y, dout, done_flag, wide);
    else
      hipLaunchKernelGGL((rba::k_hx_implicit_lds<S, NT, false>), dim3(n_hx_chunks_), dim3(NT), ylds_bytes, stream_, prm_, it,
                         d_hx_chunks_.get(), hx_win_, xin, y, dout, done_flag, wide);
  }

  void right_multiply(const void* x, void* y) override {
    use_device();
    if (!landmark_damping_valid_) run_stage2(S(0));  // the operator needs the damping records (lambda = 0)
    d_vin_.upload(static_cast<const S*>(x), nvec_, stream_);
    d_tmp_.zero(stream_);
    launch_hx(d_vin_.get(), d_tmp_.get());
    if (!sc_) all_reduce(d_tmp_.get(), nvec_);  // (SC backend: the summed matrix is on every rank)
    hipLaunchKernelGGL((rba::k_axpy_lambda<S>), dim3((nvec_ + 255) / 256), dim3(256), 0, stream_,
                       d_vin_.get(), d_tmp_.get(), sc_ ? S(0) : pose_damping_, nvec_);
    d_tmp_.download(static_cast<S*>(y), nvec_, stream_);
    sync();
  }

  // y = (S + lambda I) x through the explicitly assembled reduced matrix (tests)
  void right_multiply_explicit(const void* x, void* y) override {
    if (!ex_ready_) throw HipError{"the explicit reduced matrix is not enabled for this configuration", RBA_ERR_UNSUPPORTED};
    use_device();
    if (!landmark_damping_valid_) throw HipError{"right_multiply_explicit needs a stage 2 first", RBA_ERR_INVALID_ARGUMENT};
    if (!ex_valid_) assemble_explicit();
    d_vin_.upload(static_cast<const S*>(x), nvec_, stream_);
    // the SpMV of the fused PCG (kernels_pcg.hpp), refresh-product mode, on a cleared state
    HIP_CHECK(hipMemsetAsync(d_cg_.get(), 0, sizeof(rba::CgState), stream_));
    hipLaunchKernelGGL(rba::k_pcgs_begin, dim3(1), dim3(1), 0, stream_, d_cg_.get(), double(pose_damping_), 0);
    launch_pcgs_product(d_vin_.get(), /*period=*/1);
    hipLaunchKernelGGL((rba::k_pcgs_collect<S>), dim3((nvec_ + 255) / 256), dim3(256), 0, stream_, d_tmp_.get(),
                       q_pieces(), nvec_);
    d_tmp_.download(static_cast<S*>(y), nvec_, stream_);
    sync();
  }

  // q = M x + lambda x   (k_pcgs_spmv, refresh-product mode; lambda from the device state)
  void launch_pcgs_product(const S* x, int period) {
    launch_spmv<1>(nullptr, nullptr, nullptr, x, nullptr, nullptr, nullptr, 0.0, 0, 0, period, nullptr);
  }
  // k_pcgs_spmv<0>: direction update + product + p.q partials (also the termination test of the previous iteration)
  void launch_pcgs_direction_product() {
    constexpr int NB = rba::kPcgBlocks;
    double* part_rho = d_pcg_partials_.get();
    launch_spmv<0>(d_z_.get(), d_p_.get(), d_p2_.get(), nullptr, part_rho, part_rho + 2 * NB, d_pcgs_pq_.get(), opt_.eta,
                   opt_.min_cg_it, opt_.max_cg_it, kPcgPeriod, h_progress_);
  }

  static constexpr int kPcgPeriod = 10;  // residual_reset_period (conjugate_gradient.hpp:86-88)
  static constexpr int kPcgBlock = 5;    // iterations per captured launch graph
  static constexpr int kPcgRunAhead = 6; // single iterations the host may queue ahead of the device

  // one PCG iteration of the fused path: product + update (+ the residual refresh)
  void enqueue_pcgs_iteration(bool with_refresh) {
    constexpr int NB = rba::kPcgBlocks;
    rba::CgState* st = d_cg_.get();
    double* part_rho = d_pcg_partials_.get();
    double* part_q = part_rho + 2 * NB;
    launch_pcgs_direction_product();
    hipLaunchKernelGGL((rba::k_pcgs_update<S>), dim3(NB), dim3(256), 0, stream_, d_inv_.get(), prm_.b,
                       d_x_.get(), d_r_.get(), d_z_.get(), d_p_.get(), d_p2_.get(), q_pieces(), n_items_, n_cams_, st,
                       d_pcgs_pq_.get(), part_rho, part_q, 0, kPcgPeriod, h_progress_, 0, S(0),
                       static_cast<S*>(nullptr), series_t());
    if (!with_refresh) enqueue_series();
    if (with_refresh) {
      // residual refresh r = b - H x (conjugate_gradient.hpp:230-235)
      launch_pcgs_product(d_x_.get(), kPcgPeriod);
      hipLaunchKernelGGL((rba::k_pcgs_update<S>), dim3(NB), dim3(256), 0, stream_, d_inv_.get(), prm_.b,
                         d_x_.get(), d_r_.get(), d_z_.get(), d_p_.get(), d_p2_.get(), q_pieces(), n_items_, n_cams_,
                         st, d_pcgs_pq_.get(), part_rho, part_q, 1, kPcgPeriod, h_progress_, 0, S(0),
                         static_cast<S*>(nullptr), series_t());
      enqueue_series();
    }
  }
  // Power-series preconditioner on the fused path: the terms 1..m of the series behind the kernel that formed
  // z = t = Hpp^-1 r (k_pcgs_update / k_pcg_a1), two launches per term; the last one leaves the partials of rho.
  bool series_fused() const { return opt_.preconditioner_type == 2; }
  // the series' products through a float copy of the assembled matrix: float solvers (the copy is as accurate as their
  // vectors), square-root solver (the explicit-SC backend's matrix is in the solver's scalar already); RBA_SERIES_F32=0:
  // through the double matrix as in rounds 3-5
  bool series_f32() const { return sizeof(S) == 4 && !sc_ && opt_.preconditioner_type == 2 && env_.series_f32 != 0; }
  S* series_t() { return series_fused() ? d_pw_t_.get() : static_cast<S*>(nullptr); }
  // workgroups of a series step: a tile of 28 cameras each. The LAST term leaves the kPcgBlocks partials of rho the next
  // kernels sum (one per workgroup: that many workgroups, whatever the number of tiles); the others have no such
  // limit - on final-13682 (489 tiles) 64 workgroups walked eight tiles each, one dependent chain of loads after the
  // other, on a quarter of the compute units
  int series_step_grid(bool last) const {
    const int n_tiles = (n_cams_ + 27) / 28;
    return last ? rba::kPcgBlocks : std::max(rba::kPcgBlocks, std::min(n_tiles, 4096));
  }
  void enqueue_series() {
    if (!series_fused()) return;
    for (int i = 1; i <= opt_.power_order; ++i) {
      launch_spmv<2>(nullptr, nullptr, nullptr, d_pw_t_.get(), nullptr, nullptr, nullptr, -1.0, 0, 0, 1, nullptr, true);
      hipLaunchKernelGGL((rba::k_pcgs_series_step<S>), dim3(series_step_grid(i == opt_.power_order)), dim3(256), 0, stream_,
                         d_inv_.get(), q_pieces(), d_pw_t_.get(), d_z_.get(), d_r_.get(), n_cams_, d_cg_.get(),
                         i == opt_.power_order ? 1 : 0, d_pcg_partials_.get());
    }
  }

  // The same terms in the protocol of the MATRIX-FREE products (direction kernel, product, k_pcgs_update): behind the
  // kernel that left z = t = Hpp^-1 r. Through the assembled matrix where one is valid for this damping - the iterations
  // of a run whose products are split over the ranks, and the repeat of a solve whose assembled operator lost
  // definiteness: the PRODUCT is matrix-free again (p.q = |A p|^2 + lambda |p|^2 cannot turn negative) but an approximate
  // inverse tolerates the eps |S| error of the matrix, and order m costs m SpMVs instead of m matrix-free E0 products
  // (final-13682, order 10: 2 ms instead of 25 ms per PCG iteration) - else with matrix-free E0 products
  // (PowerSCPreconditioner::solve_assign, preconditioner.hpp:180-192: t <- Hpp^-1 E0 t, z += t), all-reduced per term.
  void enqueue_series_terms(S lambda, const int* done) {
    if (!series_fused()) return;
    const int n = nvec_;
    const bool on_matrix = ex_active_ || (explicit_off_for_solve_ && ex_ready_ && ex_valid_);
    S* t = d_pw_t_.get();
    S* e = d_pw_e_.get();
    for (int i = 1; i <= opt_.power_order; ++i) {
      const int last = i == opt_.power_order ? 1 : 0;
      if (on_matrix) {
        launch_spmv<2>(nullptr, nullptr, nullptr, t, nullptr, nullptr, nullptr, double(lambda), 0, 0, 1, nullptr, true);
        hipLaunchKernelGGL((rba::k_pcgs_series_step<S>), dim3(series_step_grid(last)), dim3(256), 0, stream_, d_inv_.get(),
                           q_pieces(), t, d_z_.get(), d_r_.get(), n_cams_, d_cg_.get(), last, d_pcg_partials_.get(), 0);
      } else {
        HIP_CHECK(hipMemsetAsync(e, 0, size_t(n) * sizeof(S), stream_));
        launch_e0(t, e, done);
        all_reduce(e, n);
        rba::QPieces<S> qp{};
        qp.qmain = e;
        hipLaunchKernelGGL((rba::k_pcgs_series_step<S>), dim3(series_step_grid(last)), dim3(256), 0, stream_, d_inv_.get(), qp,
                           t, d_z_.get(), d_r_.get(), n_cams_, d_cg_.get(), last, d_pcg_partials_.get(), 1);
      }
    }
  }

  // the termination test of the last iteration lives in the next product's prologue
  void enqueue_pcgs_final_test() { launch_pcgs_direction_product(); }

  // Launch graphs of kPcgBlock iterations (the host cannot issue two launches per 13 us iteration
  // eagerly): [0] plain, [1] with the residual refresh after the block's last iteration. All
  // per-solve parameters live in the device state, so the graphs are captured once.
  void destroy_pcg_graphs() {
    for (auto& g : pcg_graph_exec_) {
      if (g) (void)hipGraphExecDestroy(g);
      g = nullptr;
    }
  }
  void build_pcg_graphs() {
    for (int v = 0; v < 2; ++v) {
      hipGraph_t graph = nullptr;
      HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < kPcgBlock; ++i) enqueue_pcgs_iteration(v == 1 && i == kPcgBlock - 1);
      HIP_CHECK(hipStreamEndCapture(stream_, &graph));
      HIP_CHECK(hipGraphInstantiate(&pcg_graph_exec_[v], graph, nullptr, nullptr, 0));
      (void)hipGraphDestroy(graph);
    }
  }

  // The PCG from iteration `it_start` on, on the assembled matrix M, two launches per iteration
  // (kernels_pcg.hpp). State (x, r, p, rho/Q history, iteration counter) is taken over from the
  // round-1 kernels when a solve switches operators mid-way.
  void pcg_fused(S lambda, int it_start) {
    const int n = nvec_, max_it = opt_.max_cg_it;
    constexpr int NB = rba::kPcgBlocks, T = rba::kPcgThreads;
    static_assert(kPcgPeriod % kPcgBlock == 0, "graph blocks must tile the refresh period");
    rba::CgState* st = d_cg_.get();
    double* part_rho = d_pcg_partials_.get();
    // the device reads the direction of `it` completed iterations from P[(it + pswap) & 1]; it is in d_p_ now
    hipLaunchKernelGGL(rba::k_pcgs_begin, dim3(1), dim3(1), 0, stream_, st, double(lambda), (it_start - 1) & 1);
    if (it_start > 1) {
      // operator switch inside a running solve: the residual is recomputed with the operator used
      // from here on, r = b - (S + lambda I) x, exactly like the periodic refresh
      launch_pcgs_product(d_x_.get(), 1);
      hipLaunchKernelGGL((rba::k_pcgs_residual<S>), dim3((n + 255) / 256), dim3(256), 0, stream_, prm_.b,
                         d_r_.get(), q_pieces(), n, st);
    }
    hipLaunchKernelGGL((rba::k_pcg_a1<S>), dim3(NB), dim3(T), 0, stream_, d_inv_.get(), d_r_.get(), d_z_.get(),
                       n, st, part_rho);
    if (series_fused()) {
      HIP_CHECK(hipMemcpyAsync(d_pw_t_.get(), d_z_.get(), n * sizeof(S), hipMemcpyDeviceToDevice, stream_));
      enqueue_series();
    }
    if (!pcg_graph_exec_[0]) build_pcg_graphs();
    volatile int* hp = h_progress_;
    hp[0] = it_start - 1;
    hp[1] = 0;
    // run-ahead throttle: the device publishes the iteration it has started; launches queued
    // after the termination are no-ops
    auto wait_for = [&](int it, int ahead) {
      spin_while([&] { return !hp[1] && it - hp[0] > ahead; });
      return hp[1] == 0;
    };
    int it = it_start;
    bool running = true;
    while (running && it <= max_it) {
      const bool aligned = (it - 1) % kPcgBlock == 0;
      if (aligned && it + kPcgBlock - 1 <= max_it) {
        if (!(running = wait_for(it, kPcgBlock))) break;  // at most one block queued behind the running one
        const bool refresh = (it + kPcgBlock - 1) % kPcgPeriod == 0;
        HIP_CHECK(hipGraphLaunch(pcg_graph_exec_[refresh ? 1 : 0], stream_));
        it += kPcgBlock;
      } else {
        if (!(running = wait_for(it, kPcgRunAhead))) break;
        enqueue_pcgs_iteration(it % kPcgPeriod == 0);
        ++it;
      }
    }
    if (running && wait_for(it, kPcgRunAhead)) enqueue_pcgs_final_test();
    HIP_CHECK(hipGetLastError());
  }

  // The PCG from iteration `it_start` on as ONE persistent kernel with the assembled matrix in the register files
  // (kernels_pcgp.hpp). Same hand-over of the state as pcg_fused().
  bool pcg_persistent_possible() const {
    // (two ranks that share a device - the callback transport of the tests - would each wait for workgroups the other
    //  one's resident workgroups keep off the CUs)
    return pg_ready_ && !pg_broken_ && !split_ && !cb_fn_;
  }
  void pcg_persistent(int it_start) {
    rba::PgParams<S> P{};
    P.wg = d_pg_wg_.get();
    P.lane_src = d_pg_lane_src_.get();
    P.lane_col = d_pg_lane_col_.get();
    P.stage_col = d_pg_stage_col_.get();
    P.row_info = d_pg_row_info_.get();
    P.vals = sc_ ? static_cast<const void*>(scp_.vals) : static_cast<const void*>(d_ex_vals_.get());
    P.vals_solver_scalar = sc_ ? 1 : 0;
    P.inv = d_inv_.get();
    P.b = prm_.b;
    P.x = d_x_.get();
    P.r_in = d_r_.get();
    P.p_in = d_p_.get();
    P.zg = d_pg_zg_.get();
    P.xg = d_pg_xg_.get();
    P.part_rq = d_pg_part_.get();
    P.part_pq = d_pg_part_.get() + size_t(2) * rba::kPgReplicas * pg_G_;
    P.st = d_cg_.get();
    P.host_progress = h_progress_;
    P.series = series_fused() ? opt_.power_order : 0;
    P.tag_stride = P.series + 2;
    P.tg = d_pg_tg_.get();
    const unsigned span = (unsigned(opt_.max_cg_it) + 4) * unsigned(P.tag_stride);  // tags of a solve: tag_base + iteration * stride (+ term)
    if (pg_epoch_ > 0xffffffffu - 2 * span) {
      d_pg_zg_.zero(stream_);
      d_pg_xg_.zero(stream_);
      d_pg_tg_.zero(stream_);
      d_pg_part_.zero(stream_);
      pg_epoch_ = 1;
    }
    P.tag_base = pg_epoch_;
    pg_epoch_ += span;
    P.G = pg_G_;
    P.n_cams = n_cams_;
    P.switch_operator = it_start > 1 ? 1 : 0;
    P.q_tolerance = opt_.eta;
    P.min_it = opt_.min_cg_it;
    P.max_it = opt_.max_cg_it;
    P.period = kPcgPeriod;
    volatile int* hp = h_progress_;
    hp[1] = 0;
    hp[4] = 0;
    const char* trace_path = std::getenv("RBA_PCGP_TRACE");  // debug: phase stamps of the first iterations -> text file
    DevBuf<long long> d_trace;
    if (trace_path) {
      d_trace.alloc(size_t(pg_G_) * rba::kPgTraceIts * 8);
      d_trace.zero(stream_);
      P.trace = d_trace.get();
    }
    hipLaunchKernelGGL((rba::k_pcgp<S>), dim3(pg_G_), dim3(rba::kPgThreads), rba::pgp_lds_bytes<S>(), stream_, P);
    HIP_CHECK(hipGetLastError());
    if (trace_path) {
      std::vector<long long> h(size_t(pg_G_) * rba::kPgTraceIts * 8);
      d_trace.download(h.data(), h.size(), stream_);
      HIP_CHECK(hipStreamSynchronize(stream_));
      if (FILE* f = std::fopen(trace_path, "a")) {
        std::fprintf(f, "solve G %d it_start %d\n", pg_G_, it_start);
        for (int g = 0; g < pg_G_; ++g)
          for (int i = 0; i < rba::kPgTraceIts; ++i) {
            const long long* t = &h[(size_t(g) * rba::kPgTraceIts + i) * 8];
            if (t[0] == 0) break;
            std::fprintf(f, "%d %d %lld %lld %lld %lld %lld %lld %lld %lld\n", g, i, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
          }
        std::fclose(f);
      }
    }
  }

  // ---- solve = stage 2 + preconditioner + PCG ------------------------------------
  int solve(double lambda_d, void* inc_out, rba_cg_summary* cg_out) override {
    use_device();
    const S lambda = S(lambda_d);
    pcg_state_pending_ = false;  // (a solve that threw may have left it set)
    time_begin();
    run_stage2(lambda);
    time_end(&timings_.stage2_time);
    sub_collect();

    time_begin();
    invert_preconditioner_blocks(lambda);
    time_end(&timings_.compute_preconditioner_time);

    time_begin();
    hx_event_count_ = 0;
    hx_calls_ = 0;
    rba_cg_summary cg = pcg(sc_ ? S(0) : lambda);  // SC: the damping is inside the matrix
    bool redo = pcg_used_explicit_ && !sc_ && (cg.termination_type == 2 || pcg_indefinite_ || env_.force_explicit_fallback);
    if (pcg_used_explicit_ && !sc_ && !redo && env_.verify_assembled) {
      // Diagnostic (RBA_VERIFY_ASSEMBLED=1; rounds 2-3: always on): a solve that ran on the assembled matrix is checked
      // with ONE product of the reference's operator - the Q model -x.(b + r)/2 that the stopping rule watched against its
      // true value; beyond `verify_tolerance` the solve is repeated with the reference's operator. With the FLOAT matrix of
      // round 3 (S + E, |E| ~ eps |S|) the two differed by 1e-5 ... 1e-3 for solves of up to ~150 iterations, 1.5e-2 at
      // 350, ~1e-1 at 500, and the check was a safety net worth a product and a host synchronisation per solve; with the
      // DOUBLE matrix (kernels_a64.hpp) they agree to 1e-7 ... 2e-5 on venice-1778 and final-13682 up to 500 iterations
      // (profiles/r4_assembled_solve_q_model_check.log), so it is off unless asked for.
      d_tmp_.zero(stream_);
      launch_hx_implicit(d_x_.get(), d_tmp_.get(), nullptr);
      all_reduce(d_tmp_.get(), nvec_);
      double* chk = d_partials_.get() + size_t(kReduceBlocks) * 8 + 8;
      hipLaunchKernelGGL((rba::k_solve_check<S>), dim3(1), dim3(1024), 0, stream_, d_x_.get(), prm_.b, d_r_.get(),
                         d_tmp_.get(), lambda, nvec_, chk);
      double* h = pinned_doubles(kPinCheck);
      HIP_CHECK(hipMemcpyAsync(h, chk, 4 * sizeof(double), hipMemcpyDeviceToHost, stream_));
      sync();
      ++pcg_counters_.products_matrix_free;
      const double q_rec = h[0], q_true = h[1];
      const bool ok = std::isfinite(q_true) && std::abs(q_true - q_rec) <= env_.verify_tolerance * std::abs(q_true);
      if (env_.verbose)
        std::fprintf(stderr, "[rootba_hip] assembled solve (%d iterations, lambda %.2e): Q model %.6e, true %.6e, |r| %.3e "
                     "true %.3e -> %s\n", cg.num_iterations, double(lambda), -0.5 * q_rec, -0.5 * q_true, std::sqrt(h[3]),
                     std::sqrt(h[2]), ok ? "kept" : "repeated matrix-free");
      redo = !ok;
    }
    if (redo) {
      // The assembled operator is S + E with |E| ~ eps |S|: unlike the square-root product
      // (p.q = |A p|^2 + lambda |p|^2 >= 0 by construction) it can lose definiteness when
      // lambda < eps |S|. The reference's operator cannot - repeat this solve matrix-free.
      ++pcg_counters_.solves_repeated_matrix_free;
      FlagScope off(explicit_off_for_solve_, true);
      cg = pcg(lambda);
    }
    if (inc_out) {
      hipLaunchKernelGGL((rba::k_negate<S>), dim3((nvec_ + 255) / 256), dim3(256), 0, stream_,
                         d_x_.get(), nvec_);
      d_x_.download(h_vec_stage_, nvec_, stream_);  // (pinned; copied to the caller's buffer after the synchronisation below)
    } else {
      // rba_lm_step: the increment stays on the device (apply(nullptr) reads d_inc_), the host gets its norm - and,
      // when nobody has read it yet, the final PCG state
      // (+ D inc for the back-substitution and, inside rba_lm_step, the failure word of the linearisation: two launches
      //  of ~5 us less)
      const bool flag_here = lin_flag_deferred_ && results_go_direct();
      hipLaunchKernelGGL((rba::k_finish_increment<S>), dim3(1), dim3(1024), 0, stream_, d_x_.get(), d_inc_.get(),
                         nvec_, pinned_doubles(kPinInc), static_cast<const rba::CgState*>(d_cg_.get()),
                         pcg_state_pending_ ? reinterpret_cast<rba::CgState*>(h_pinned_)
                                            : static_cast<rba::CgState*>(nullptr),
                         static_cast<const S*>(prm_.pose_scaling), sc_ ? static_cast<S*>(nullptr) : d_xs_.get(),
                         flag_here ? d_fail_.get() : static_cast<int*>(nullptr),
                         flag_here ? pinned_int(kPinFailLin) : static_cast<int*>(nullptr), 1);
      inc_prescaled_ = !sc_;
      if (flag_here) lin_flag_deferred_ = false;
    }
    time_end(&timings_.solve_reduced_system_time);
    solve_cg_ = cg;
    if (solve_defer_) return RBA_OK;  // rba_lm_step calls solve_collect() after its synchronisation
    if (lm_async_ || inc_out) {  // the caller reads the increment, and the product timers below need their events completed
      stamp_close();
      sync();
      flush_timers();
    }
    if (inc_out) std::memcpy(inc_out, h_vec_stage_, size_t(nvec_) * sizeof(S));
    solve_collect(cg_out);
    return RBA_OK;
  }

  // What a solve leaves for the host once its kernels have completed: the PCG summary (if the state was not read
  // inside), the product timers and, once, the break-even of the operator switch.
  void solve_collect(rba_cg_summary* cg_out) {
    rba_cg_summary cg = solve_cg_;
    if (pcg_state_pending_) {
      pcg_collect(&cg, pcg_it_first_assembled_);
      pcg_state_pending_ = false;
    }
    // H*x launches that did real work: one per PCG iteration plus the residual
    // refreshes; launches queued after termination are no-ops and are excluded
    const int real_hx = cg.num_iterations + cg.num_iterations / 10;
    double hx_ms = 0;
    int timed = 0;
    for (int i = 0; i < hx_event_count_; ++i) {
      if (hx_event_call_[i] >= real_hx) break;
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, hx_events_[2 * i], hx_events_[2 * i + 1]));
      hx_ms += ms;
      ++timed;
    }
    timings_.hx_time = hx_ms * 1e-3;  // sum over the TIMED products
    timings_.hx_calls = timed;
    // (RBA_DETERMINISTIC=1: the switch stays at its initial product count - a measured break-even differs from run
    //  to run, and with it the iteration from which a solve continues on the assembled matrix)
    if (asm_pending_ && env_.deterministic) {
      asm_pending_ = false;
      asm_measured_ = true;
    }
    if (asm_pending_ && timed > 0) {
      // break-even of the switch (ski rental): as many matrix-free products as one assembly costs.
      // Identical on all ranks? No - timings differ, so rank 0's value is broadcast (max is enough:
      // every rank must switch in the same iteration).
      float asm_ms = 0;
      HIP_CHECK(hipEventElapsedTime(&asm_ms, ev_asm0_, ev_asm1_));
      // Break-even with BOTH products: an assembly pays once the iterations that follow it have
      // saved its cost, asm / (t_matrix-free - t_assembled) of them. Rounds 2-5 divided by the matrix-free time alone -
      // right for a banded matrix in the register files (7.7 us against 133 per iteration), wrong by an order of
      // magnitude for a nearly dense one that streams from HBM (venice-1778+tail: 137 us against 143 - two assemblies of
      // 4.5 GB each for 810 products that were 4 % cheaper). t_assembled: measured on this very solve where the stage
      // timers run (its time less the assembly and the matrix-free products), else priced from the bytes of a product
      // at the 3.8 TB/s the streaming SpMV reaches (the persistent kernel: 8 us).
      const double t_mf = hx_ms / timed;
      double t_as = pcg_last_was_persistent_ ? 0.008 : 0.007 + double(ex_nnz_) * 81 * 8 / 3.8e9;
      if (last_as_products_ >= 8 && timings_.solve_reduced_system_time > 0) {
        const double rest = timings_.solve_reduced_system_time * 1e3 - double(asm_ms) - double(last_mf_products_) * t_mf;
        if (rest > 0) t_as = rest / double(last_as_products_);
      }
      const double be = t_mf > 1.02 * t_as ? double(asm_ms) / (t_mf - t_as) : 1e9;
      int t = int(std::min(be, 1e6) + 0.5);
      t = std::max(2, std::min(opt_.max_cg_it + 1, t));
      measured_t_as_ms_ = t_as;
      if (comm_ || cb_fn_) {
        d_scratch_int_.upload(&t, 1, stream_);
        all_reduce(d_scratch_int_.get(), 1, kNcclMax);
        d_scratch_int_.download(&t, 1, stream_);
        sync();
      }
      if (env_.verbose)
        std::fprintf(stderr, "[rootba_hip] assembly %.3f ms, matrix-free product %.3f ms, iteration on the assembled matrix %.4f ms "
                     "-> explicit_after = %d\n", double(asm_ms), t_mf, measured_t_as_ms_, t);
      explicit_after_ = t;
      // the early exit from the rental (a solve whose stopping quantity rises at iteration 4 is taken for a long one,
      // 40+ iterations): only where 40 cheaper iterations pay for an assembly
      early_switch_pays_ = t <= 40;
      asm_pending_ = false;
      asm_measured_ = true;
    }
    if (cg_out) *cg_out = cg;
  }

  rba_cg_summary pcg(S lambda) {
    rba_cg_summary summary{0, 0};
    const int n = nvec_;
    rba::CgState* st = d_cg_.get();
    const int* done = &st->done;
    const S* b = prm_.b;
    constexpr int NB = rba::kPcgBlocks;
    double* part_rho = d_pcg_partials_.get();
    double* part_pq = part_rho + NB;
    double* part_q1 = part_pq + NB;
    const int max_it = opt_.max_cg_it, min_it = opt_.min_cg_it;
    const double eta = opt_.eta;
    rba::CgState* hst = reinterpret_cast<rba::CgState*>(h_pinned_);
    // (since round 6 also the solves with the power-series preconditioner whose products are matrix-free - no assembled
    //  matrix, or the repeat of a solve whose assembled operator broke down: rounds 1-5 kept the seven-kernel loop of round
    //  1 for them)
    const bool mf_protocol = !sc_;
    if (mf_protocol) {
      unsigned long long* const stamp_ptr_ = stamp();
      hipLaunchKernelGGL((rba::k_pcgs_start<S>), dim3(NB), dim3(256), 0, stream_, d_inv_.get(), b, d_x_.get(),
                         d_r_.get(), d_z_.get(), n_cams_, st, part_rho, double(lambda), 0, stamp_ptr_);
    }
    else {
      unsigned long long* const stamp_ptr_ = stamp();
      hipLaunchKernelGGL((rba::k_pcg_init<S>), dim3(1), dim3(1024), 0, stream_, b, d_x_.get(), d_r_.get(), n, st,
                         stamp_ptr_);
    }
    // The host polls the device state lazily: every iteration at first (many
    // solves need 2-3 iterations), every 4th later; kernels queued past the end
    // are no-ops (`done`).
    ex_active_ = false;
    pcg_used_explicit_ = false;
    // (the power series runs on the fused path too, through the assembled matrix - not in the repeat of a solve whose
    //  products went back to matrix-free)
    const bool fused = n_items_ > 0 && (opt_.preconditioner_type != 2 || !explicit_off_for_solve_) && !split_;
    bool go_fused = sc_ && fused;  // explicit Schur-complement backend: the matrix exists from the start
    int it = 1, it_first_assembled = 1;
    // Block-diagonal preconditioners: the matrix-free iterations run in the protocol of the fused PCG (kernels_pcg.hpp) -
    // direction kernel (test of the previous iteration, rho, beta, p, D p), product, update kernel (p.q, alpha, x, r,
    // z = M^-1 r, partials) - and the host follows through the pinned progress words instead of copies of the state.
    bool mf_open = false;  // iterations were enqueued whose closing test still lives in the next prologue
    if (mf_protocol) {
      volatile int* hp = h_progress_;
      hp[0] = 0;
      hp[1] = 0;
      hp[2] = hp[3] = 0x7fc00000;  // zeta of iterations 3, 4 (float bits): NaN until the device publishes them
      auto direction = [&](bool test_only = false) {
        const bool pre = !ex_active_;
        hipLaunchKernelGGL((rba::k_pcgs_direction<S>), dim3(NB), dim3(256), 0, stream_, d_z_.get(), d_p_.get(),
                           d_q_.get(), n, st, part_rho, part_q1, static_cast<const S*>(pre ? prm_.pose_scaling : nullptr),
                           pre ? d_xs_.get() : static_cast<S*>(nullptr), eta, min_it, max_it, h_progress_,
                           test_only ? 1 : 0);
        return pre;
      };
      auto update = [&](int phase, const S* product) {
        rba::QPieces<S> qp{};  // (matrix-free product: one complete vector)
        qp.qmain = product;
        hipLaunchKernelGGL((rba::k_pcgs_update<S>), dim3(NB), dim3(256), 0, stream_, d_inv_.get(), b, d_x_.get(),
                           d_r_.get(), d_z_.get(), d_p_.get(), d_p_.get(), qp, 0, n_cams_, st,
                           static_cast<const double*>(nullptr), part_rho, part_q1, phase, kPcgPeriod, h_progress_, 1,
                           lambda, d_tmp_.get(), series_t());
        enqueue_series_terms(lambda, done);  // (power series: z = Hpp^-1 r was only its first term)
      };
      // the device publishes the iteration it has started (hp[0]) or the end of the solve (hp[1])
      auto started = [&](int k) {
        spin_while([&] { return !hp[1] && hp[0] < k; });
        return hp[1] == 0;
      };
      bool running = true;
      for (; it <= max_it; ++it) {
        // Early switch (measured break-even only): solves are bimodal - they end after two or three iterations or run
        // for tens to hundreds - and the long ones show it early: the stopping quantity zeta_i = i (Q_i - Q_i-1) / Q_i
        // (conjugate_gradient.hpp:263-276) RISES from iteration 3 to 4 where it falls steadily in the solves that end
        // within a dozen iterations (ladybug / trafalgar / venice: every solve of 40+ iterations rises, those of 6 - 24
        // fall). A solve that rises is switched to the assembled matrix at iteration 5 instead of after
        // `explicit_after_` (~12) products: eight products of 0.14 ms saved per long solve on venice. All ranks of a
        // sharded run see the same zeta (identical scalars by construction) and decide alike.
        bool tested = false, switch_now = false;
        if (it == 5 && explicit_auto_ && early_switch_pays_ && ex_ready_ && !explicit_off_for_solve_ && !ex_valid_ &&
            explicit_after_ >= it) {
          direction(true);
          if (!(running = started(it))) break;
          tested = true;
          float z3, z4;
          std::atomic_thread_fence(std::memory_order_acquire);  // (pairs with the release store of the progress word)
          const int b3 = hp[2], b4 = hp[3];
          std::memcpy(&z3, &b3, sizeof z3);
          std::memcpy(&z4, &b4, sizeof z4);
          switch_now = std::isfinite(z3) && std::isfinite(z4) && z4 >= z3;
          if (switch_now) ++pcg_counters_.early_switches;
        }
        // (power series: every iteration costs 1 + power_order products - the assembly pays off at once)
        if (ex_ready_ && !ex_active_ && !explicit_off_for_solve_ &&
            (it > (series_fused() ? 0 : explicit_after_) || switch_now)) {
          // (the verdict on the iterations so far first: a solve that ends exactly here must not pay for an assembly)
          if (it >= 3 && !ex_valid_ && !tested) {
            direction(true);
            if (!(running = started(it))) break;
          }
          if (!ex_valid_) assemble_explicit();
          ex_active_ = true;
          pcg_used_explicit_ = true;
          it_first_assembled = it;
          if (fused) {
            go_fused = true;
            break;
          }
        }
        if (it == 1 && series_fused()) {
          // the series behind the first z = Hpp^-1 b (k_pcgs_start) - here, behind the operator decision of iteration 1:
          // a solve that goes to the assembled matrix at once never pays for matrix-free terms
          HIP_CHECK(hipMemcpyAsync(d_pw_t_.get(), d_z_.get(), size_t(n) * sizeof(S), hipMemcpyDeviceToDevice, stream_));
          enqueue_series_terms(lambda, done);
        }
        operand_prescaled_ = direction();
        // many solves need two or three iterations: products 3 to 8 wait for the verdict of the test that precedes
        // them, later ones are queued ahead (launches behind the end of the solve are no-ops). The first two never
        // wait: the Q-model test cannot end a solve after one iteration (zeta = 1 * (Q_1 - 0) / Q_1 = 1 > eta).
        // (more than one rank: every product is followed by a collective that all ranks enter whether the solve has ended
        //  or not - no product is queued ahead of its verdict there)
        if (it >= 3 && (it <= 8 || it % 4 == 0 || comm_ || cb_fn_) && !(running = started(it))) {
          operand_prescaled_ = false;
          break;
        }
        launch_hx(d_p_.get(), d_q_.get(), done);
        operand_prescaled_ = false;
        if ((!sc_ && !ex_active_) || split_) all_reduce(d_q_.get(), n);
        update(0, d_q_.get());
        if (it % kPcgPeriod == 0) {
          // residual refresh r = b - H x (conjugate_gradient.hpp:230-235)
          launch_hx(d_x_.get(), d_tmp_.get(), done);
          if ((!sc_ && !ex_active_) || split_) all_reduce(d_tmp_.get(), n);
          update(1, d_tmp_.get());
        }
        mf_open = true;
      }
      if (!go_fused) {
        if (running && mf_open) direction();  // the test of the last iteration
        if (solve_defer_) {
          // rba_lm_step: the state travels with the end of the solve (k_finish_increment) and is read at the
          // iteration's one synchronisation (solve_collect)
          pcg_state_pending_ = true;
          pcg_it_first_assembled_ = it_first_assembled;
          return summary;
        }
        HIP_CHECK(hipMemcpyAsync(hst, st, sizeof(rba::CgState), hipMemcpyDeviceToHost, stream_));
        sync();
      }
    }
    if (!mf_protocol && !go_fused)
      throw HipError{"the explicit Schur-complement backend has no matrix to iterate on", RBA_ERR_INVALID_ARGUMENT};
    if (go_fused) {
      bool persistent = pcg_persistent_possible();
      if (persistent) {
        // (explicit-SC backend: the state comes from k_pcg_init, the damping is inside the matrix)
        // (and the power series: its solves come here from k_pcg_init as well)
        if (sc_ || series_fused()) hipLaunchKernelGGL(rba::k_pcgs_begin, dim3(1), dim3(1), 0, stream_, st, double(lambda), 0);
        {
          // the solve's entry state (x, the PCG scalars; r, z, p are inputs the kernel never writes): restored before the
          // two-launch path takes over when the kernel - this rank's or, in a sharded run, any rank's - gave up. Always
          // saved: a workgroup that finished before another one's bounded wait expired has written x and `done`
          // 
          if (!d_pg_xsave_.get()) {
            d_pg_xsave_.alloc(size_t(n));
            d_pg_stsave_.alloc(1);
          }
          HIP_CHECK(hipMemcpyAsync(d_pg_xsave_.get(), d_x_.get(), size_t(n) * sizeof(S), hipMemcpyDeviceToDevice, stream_));
          HIP_CHECK(hipMemcpyAsync(d_pg_stsave_.get(), st, sizeof(rba::CgState), hipMemcpyDeviceToDevice, stream_));
        }
        pcg_persistent(it);
        HIP_CHECK(hipMemcpyAsync(hst, st, sizeof(rba::CgState), hipMemcpyDeviceToHost, stream_));
        sync();
        const bool finished = h_progress_[4] == 0;
        // (RBA_PCGP_TEST_GIVE_UP=1, sharded runs: behave as if this rank's kernel had given up - the test of the path below)
        const bool mine_gave_up = !finished || (env_.pcgp_test_give_up && comm_);
        int gave_up = mine_gave_up ? 1 : 0;
        if (comm_) {
          // sharded run: the ranks hold replicas of this solve and must continue alike - one rank falling back to the
          // two-launch path on its own would round differently, and a different iteration count on one rank is a
          // mismatched collective in the next solve. All fall back if any gave up.
          d_scratch_int_.upload(&gave_up, 1, stream_);
          all_reduce(d_scratch_int_.get(), 1, kNcclMax);
          d_scratch_int_.download(&gave_up, 1, stream_);
          sync();
        }
        if (gave_up) {
          // back to the entry state - whether this rank's kernel finished (another rank's did not) or some of its
          // workgroups had already written the end of the solve when another one gave up -, then everybody takes the
          // two-launch path
          HIP_CHECK(hipMemcpyAsync(d_x_.get(), d_pg_xsave_.get(), size_t(n) * sizeof(S), hipMemcpyDeviceToDevice, stream_));
          HIP_CHECK(hipMemcpyAsync(st, d_pg_stsave_.get(), sizeof(rba::CgState), hipMemcpyDeviceToDevice, stream_));
          // a workgroup waited ~1 s for a granule (not every workgroup resident? another process on the device?): nothing
          // of the state was written - continue in two launches per iteration and keep to them
          std::fprintf(stderr, "[rootba_hip] persistent PCG gave up waiting (%d workgroups); two-launch path from now on\n", pg_G_);
          pg_broken_ = true;
          persistent = false;
        } else {
          ++pcg_counters_.solves_persistent;
          pcg_last_persistent_ = true;
        }
      }
      if (!persistent) {
        pcg_fused(lambda, it);
        HIP_CHECK(hipMemcpyAsync(hst, st, sizeof(rba::CgState), hipMemcpyDeviceToHost, stream_));
        sync();
      }
    }
    ex_active_ = false;
    pcg_collect(&summary, it_first_assembled);
    pcg_last_persistent_ = false;
    return summary;
  }

  // summary and counters of a finished PCG from the copy of its state in the pinned page
  void pcg_collect(rba_cg_summary* out, int it_first_assembled) {
    const rba::CgState* hst = reinterpret_cast<const rba::CgState*>(h_pinned_);
    rba_cg_summary summary{0, 0};
    summary.termination_type = hst->termination;
    summary.num_iterations = hst->result_iter;
    pcg_indefinite_ = hst->indefinite != 0;
    {
      // products that did real work: one per iteration + one per residual refresh (every 10th), split at the
      // iteration `it_switch` from which the assembled matrix was used (+ the refresh product of the switch)
      const int n_it = summary.num_iterations;
      const int it_switch = pcg_used_explicit_ && !sc_ ? std::min(it_first_assembled, n_it + 1) : (sc_ ? 1 : n_it + 1);
      const int64_t mf_it = it_switch - 1, as_it = n_it - mf_it;
      const int64_t series = opt_.preconditioner_type == 2 ? opt_.power_order : 0;  // products of the preconditioner
      pcg_counters_.iterations += n_it;
      pcg_counters_.products_matrix_free += mf_it + mf_it / 10 + (pcg_used_explicit_ || sc_ ? 0 : series * mf_it);
      const int64_t as_products = as_it > 0 ? as_it + (n_it / 10 - mf_it / 10) + (it_switch > 1 ? 1 : 0) + series * as_it : 0;
      pcg_counters_.products_assembled += as_products;
      last_mf_products_ = mf_it + mf_it / 10;
      last_as_products_ = as_products;
      pcg_last_was_persistent_ = pcg_last_persistent_;
      // (persistent kernel: the matrix is read ONCE per solve and multiplied out of the register files - these products
      //  move no matrix bytes; rba_byte_model.persistent_solve / .persistent_iteration price such a solve)
      if (pcg_last_persistent_) {
        pcg_counters_.products_assembled_resident += as_products;
        pcg_counters_.iterations_resident += as_it > 0 ? as_it : 0;
      }
    }
    *out = summary;
  }

  // ---- apply ------------------------------------------------------------------------
  int apply(const void* inc, double* l_diff_out, bool update_cams) override {
    if (!in_lm_step_) lm_.ri_is_current = false;  // (a caller changes the state between two rba_lm_step calls: the cached cost is another state's)
    use_device();
    if (!landmark_damping_valid_) run_stage2(S(0));
    time_begin();
    if (inc) {  // (nullptr: left there by the solve of rba_lm_step)
      sync();  // (the pinned stage is free: nothing queued reads it)
      std::memcpy(h_vec_stage_, inc, size_t(nvec_) * sizeof(S));
      d_inc_.upload(h_vec_stage_, nvec_, stream_);
    }
    if (sc_) {
      hipLaunchKernelGGL((rba::k_sc_back_substitute<S>), dim3((n_lms_ + 255) / 256), dim3(256), 0, stream_,
                         scp_, d_inc_.get());
    } else {
      // tiled landmarks (k <= 32, implicit-Q configuration): one lane-per-row pass; the rest: two passes
      int lm0 = 0;
      int64_t o0 = 0;
      // D inc for the unscaled Jacobian rows (inside rba_lm_step the end of the solve has left it in d_xs_ already)
      const S* xin = (inc_prescaled_ && !inc) ? d_xs_.get() : scaled_operand(d_inc_.get());
      if (n_tiles_ > 0) {
        {  // (evaluated once, ahead of the launch macro)
          const rba::Params<S> prm_st_ = prm_stamped();
          hipLaunchKernelGGL((rba::k_bs_tile<S>), dim3((n_tiles_ + 3) / 4), dim3(256), 0, stream_, prm_st_,
                             implicit_tiles(), xin);
        }
        lm0 = imp_end_[4];
        o0 = n_obs_tiled_;
      }
      if (o0 < n_obs_) {
        hipLaunchKernelGGL((rba::k_bs_obs<S>), dim3(unsigned((n_obs_ - o0 + 255) / 256)), dim3(256), 0, stream_, prm_,
                           xin, o0, int64_t(n_obs_));
        if (big_begin_ > lm0 && lm0 > 0)  // only 32 < k <= 112 left: a wavefront per landmark
          hipLaunchKernelGGL((rba::k_bs_landmark_wave<S>), dim3((big_begin_ - lm0 + 3) / 4), dim3(256), 0, stream_,
                             prm_, lm0, big_begin_);
        else if (big_begin_ > lm0)
          hipLaunchKernelGGL((rba::k_bs_landmark<S>), dim3((big_begin_ - lm0 + 255) / 256), dim3(256), 0, stream_,
                             prm_, lm0, big_begin_);
        if (n_big_ > 0)
          hipLaunchKernelGGL((rba::k_bs_landmark_big<S>), dim3(n_big_), dim3(256), 0, stream_, prm_, big_begin_);
      }
    }
    inc_prescaled_ = false;
    const int blocks = std::min(kReduceBlocks, (n_lms_ + 255) / 256);
    hipLaunchKernelGGL((rba::k_sum_ldiff), dim3(blocks), dim3(256), 0, stream_,
                       d_lm_ldiff_.get(), n_lms_, d_partials_.get());
    double* red = d_partials_.get() + size_t(kReduceBlocks) * 8;
    double* l_diff = pinned_doubles(kPinLdiff);
    int* fail = pinned_int(kPinFailApply);
    if (results_go_direct()) {
      // l_diff and the failure word (bits 2: back-substitution, 4: block inversion of the solve) land in the pinned page
      hipLaunchKernelGGL((rba::k_reduce_rows<1>), dim3(1), dim3(256), 0, stream_, d_partials_.get(), int64_t(blocks),
                         red, l_diff, d_fail_.get(), fail, 2 | 4, static_cast<unsigned long long*>(nullptr),
                         static_cast<double*>(nullptr));
    } else if (lm_merge_end_) {
      // rba_lm_step of a sharded run: l_diff and the failure bits (the deferred one of the linearisation included) join
      // the cost sums of the trial point - ONE all-reduce ends the iteration (compute_error_enqueue)
      hipLaunchKernelGGL((rba::k_reduce_rows<1>), dim3(1), dim3(256), 0, stream_, d_partials_.get(), int64_t(blocks),
                         d_endred_.get() + 8, static_cast<double*>(nullptr), d_fail_.get(), static_cast<int*>(nullptr),
                         1 | 2 | 4, static_cast<unsigned long long*>(nullptr), d_endred_.get() + 9);
      lin_flag_deferred_ = false;
    } else {
      hipLaunchKernelGGL((rba::k_reduce_rows<1>), dim3(1), dim3(256), 0, stream_, d_partials_.get(), int64_t(blocks),
                         red, static_cast<double*>(nullptr), static_cast<int*>(nullptr), static_cast<int*>(nullptr), 0,
                         static_cast<unsigned long long*>(nullptr), static_cast<double*>(nullptr));
      all_reduce(red, 1);
      all_reduce(d_fail_.get(), 1, kNcclMax);
      HIP_CHECK(hipMemcpyAsync(l_diff, red, sizeof(double), hipMemcpyDeviceToHost, stream_));
      HIP_CHECK(hipMemcpyAsync(fail, d_fail_.get(), sizeof(int), hipMemcpyDeviceToHost, stream_));
      hipLaunchKernelGGL(rba::k_publish_flag, dim3(1), dim3(1), 0, stream_, d_fail_.get(), d_scratch_int_.get(), 2 | 4);
    }
    time_end(&timings_.back_substitution_time);
    // the back-substitution leaves the blocks undamped in the reference
    // (ipp:247-248); here damped rows are rebuilt by the next stage 2 anyway
    // (inside rba_lm_step the outcome is read at the iteration's last synchronisation, apply_outcome(): the camera
    //  update below is enqueued regardless - a failed step is restored from the backup by the LM loop)
    if (!lm_async_ && apply_outcome(l_diff_out) != RBA_OK) return RBA_NUMERICAL_FAILURE;
    if (mixed_) {
      // the float landmark increments go to the double masters; the float state is re-rounded from them
      const int64_t nl = 3 * int64_t(n_lms_);
      hipLaunchKernelGGL(rba::k_mixed_update_landmarks, dim3(unsigned((nl + 255) / 256)), dim3(256), 0, stream_,
                         d_lms64_.get(), reinterpret_cast<float*>(d_lms_.get()),
                         reinterpret_cast<const float*>(d_lm_inc_.get()), nl);
    }
    if (update_cams) {
      time_begin();
      if (mixed_) {
        unsigned long long* const stamp_ptr_ = stamp();
        hipLaunchKernelGGL(rba::k_mixed_update_cameras, dim3((n_cams_ + 63) / 64), dim3(64), 0, stream_,
                           d_cams64_.get(), reinterpret_cast<float*>(d_cams_.get()),
                           reinterpret_cast<const float*>(d_inc_.get()),
                           reinterpret_cast<const float*>(prm_.pose_scaling), n_cams_, stamp_ptr_);
      }
      else {
        const rba::Params<S> prm_st_ = prm_stamped();
        hipLaunchKernelGGL((rba::k_update_cameras<S>), dim3((n_cams_ + 63) / 64), dim3(64), 0,
                           stream_, prm_st_, d_inc_.get(), cams_backup_in_update_ ? d_cams_bak_.get() : static_cast<S*>(nullptr));
      }
      time_end(&timings_.update_cameras_time);
    }
    return RBA_OK;
  }

  // l_diff / failure flag of the last back-substitution (pinned slots; valid after a synchronisation)
  int apply_outcome(double* l_diff_out) {
    const double l_diff = *pinned_doubles(kPinLdiff);
    if (!std::isfinite(l_diff) || (*pinned_int(kPinFailApply) & 2)) {
      *l_diff_out = std::numeric_limits<double>::quiet_NaN();
      return RBA_NUMERICAL_FAILURE;
    }
    *l_diff_out = double(S(l_diff));
    return RBA_OK;
  }

  // ---- LM driver (optimize_lm_ours) -----------------------------------------------------
  // The reference's nested loop (bal_bundle_adjustment.cpp:291-521) as a
  // resumable state machine: one call of lm_step() = one LM iteration (one row
  // of the iteration log), so callers can time / interleave iterations.
  void lm_begin() override {
    lm_ = LmState{};
    lm_.lambda = S(1.0 / opt_.initial_trust_region_radius);
    lm_.lambda_vee = S(opt_.initial_vee);
    lm_.active = true;
  }

  // returns 1 while the loop continues, 0 once terminated (row still valid
  // unless the termination was a numerical failure before any work)
  int lm_step(rba_lm_iteration* out) override {
    if (!lm_.active) lm_begin();
    const S min_lambda = S(1.0 / opt_.max_trust_region_radius);
    const S max_lambda = S(1.0 / opt_.min_trust_region_radius);
    const int max_lm_iter = opt_.max_num_iterations;
    rba_lm_iteration row{};
    row.iteration = lm_.it;
    if (lm_.terminated || lm_.it > max_lm_iter) {
      lm_.terminated = true;
      *out = row;
      return 0;
    }
    reset_timings();
    const double t_it = wall_seconds();
    // Inside this function stage results and timers are collected at the iteration's own synchronisation points (the
    // PCG's polls, the increment after the solve, the end) instead of after every stage; the unstaged sub-stage timers
    // keep the per-stage synchronisation they are read at.
    FlagScope step_scope(in_lm_step_, true);
    FlagScope async_scope(lm_async_, !sub_timing());
    FlagScope fuse_scope(lm_fuse_, !sub_timing());
    auto finish = [&](bool keep_going) {
      if (lm_async_) {
        stamp_close();
        sync();
        flush_timers();
      }
      row.iteration_time = wall_seconds() - t_it;
      row.stage1_time = timings_.stage1_time;
      row.stage2_time = timings_.stage2_time;
      row.precond_time = timings_.compute_preconditioner_time;
      row.pcg_time = timings_.solve_reduced_system_time;
      row.backsub_time = timings_.back_substitution_time;
      row.residual_time = timings_.residual_evaluation_time;
      *out = row;
      if (lm_.it > max_lm_iter) lm_.terminated = true;
      return (keep_going && !lm_.terminated) ? 1 : 0;
    };
    // the reference re-evaluates the error at every outer iteration (bal_bundle_adjustment.cpp:297-301,
    // with a TODO to avoid it); so does this loop - the metric of SURVEY.md 8d includes both evaluations
    // - except where the state has not changed since its error was evaluated: after an accepted step `lm_.ri` IS the
    // trial evaluation of the state that is now current (same kernel, same inputs, same fixed summation order: the
    // re-evaluation would return it bit for bit), so it is reused
    const bool fresh_cost = lm_.it == 0 || (lm_.need_linearize && !lm_.ri_is_current);
    if (fresh_cost) compute_error_enqueue(pinned_doubles(kPinCe0), /*side=*/lm_async_ && lm_.it > 0);
    auto cost_is_valid = [&]() {  // (after a synchronisation)
      if (fresh_cost) compute_error_parse(pinned_doubles(kPinCe0), &lm_.ri);
      return lm_.ri.is_numerically_valid != 0;
    };
    if (lm_.it == 0) {
      sync();
      if (!cost_is_valid()) {
        lm_.terminated = true;
        lm_.termination = -1;
        return finish(false);
      }
      // iteration 0 is evaluation only (bal_bundle_adjustment.cpp:311-322)
      row.cost = lm_.ri.all_error;
      row.cost_valid = lm_.ri.valid_error;
      row.num_obs = int(lm_.ri.all_num_obs);
      row.num_obs_valid = int(lm_.ri.valid_num_obs);
      row.residual_sum = lm_.ri.all_residual_sum;
      row.residual_sum_valid = lm_.ri.valid_residual_sum;
      row.lambda = lm_.lambda;
      row.step_is_successful = row.step_is_valid = 1;
      lm_.prev_all = lm_.ri.all_error;
      lm_.prev_valid = lm_.ri.valid_error;
      lm_.it = 1;
      lm_.need_linearize = true;
      lm_.ri_is_current = true;
      return finish(true);
    }
    const bool linearized = lm_.need_linearize;
    if (linearized) {
      if (linearize(nullptr) != RBA_OK) {  // (asynchronous: the failure flag is read after the solve below)
        lm_.terminated = true;
        lm_.termination = -1;
        return finish(false);
      }
      lm_.need_linearize = false;
    }
    const rba_residual_info& ri = lm_.ri;
    row.lambda = lm_.lambda;
    rba_cg_summary cg{};
    // The increment stays on the device, and inside this loop nothing about the solve is needed on the host before the
    // trial point has been evaluated: the back-substitution and the cost evaluation are queued right behind the solve
    // and ONE synchronisation delivers the PCG summary, the increment's norm and non-finite count, l_diff and the
    // costs. The two outcomes that used to stop short of the back-substitution - a non-finite state / Jacobian, a
    // non-finite increment - undo it from the backup instead.
    const bool one_sync = lm_async_;
    {
      FlagScope defer(solve_defer_, one_sync);
      solve(lm_.lambda, nullptr, &cg);
    }
    double l_diff_d = 0;
    bool applied = false;
    join_side();  // (the cost evaluation of the current state has read it: from here on it changes)
    if (one_sync) {
      // BalProblem::backup() of the step (bal_bundle_adjustment.cpp:401): the landmarks by a copy, the cameras by the
      // kernel that replaces them (k_update_cameras; mixed precision keeps its copies)
      const bool fold_cams = lm_fuse_ && !mixed_;
      if (!fold_cams)
        backup();
      else
        HIP_CHECK(hipMemcpyAsync(d_lms_bak_.get(), d_lms_.get(), d_lms_.size() * sizeof(S), hipMemcpyDeviceToDevice, stream_));
      {
        FlagScope fold(cams_backup_in_update_, fold_cams);
        FlagScope merge(lm_merge_end_, lm_fuse_ && !results_go_direct());
        lm_end_merged_ = false;
        apply(nullptr, &l_diff_d, true);
        compute_error_enqueue(pinned_doubles(kPinCe1));
      }
      stamp_close();
      sync();
      flush_timers();
      if (lm_end_merged_) {
        // sharded run: the iteration's sums arrived in one block (rba::kEndRed) - filed where the code below reads them
        const double* e = pinned_doubles(kPinEnd);
        std::memcpy(pinned_doubles(kPinCe1), e, 8 * sizeof(double));
        *pinned_doubles(kPinLdiff) = e[8];
        *pinned_int(kPinFailLin) = e[9] > 0 ? 1 : 0;
        *pinned_int(kPinFailApply) = (e[10] > 0 ? 2 : 0) | (e[11] > 0 ? 4 : 0);
      }
      solve_collect(&cg);
      applied = true;
    }
    if (!cost_is_valid() || (linearized && (*pinned_int(kPinFailLin) & 1))) {
      // non-finite residuals / Jacobians at this state (detected by the cost evaluation or the linearisation that
      // were queued ahead of the solve): numerical failure, as where the reference returns an empty vector
      if (applied) restore();
      lm_.terminated = true;
      lm_.termination = -1;
      lm_.need_linearize = true;
      return finish(false);
    }
    row.cg_iterations = cg.num_iterations;
    row.cg_termination = cg.termination_type;
    const double nrm = pinned_doubles(kPinInc)[0];
    const bool finite = pinned_doubles(kPinInc)[1] == 0.0;
    row.inc_norm = std::sqrt(nrm);
    if (!finite) {
      // non-finite increment: reject, increase damping (:360-399)
      if (applied) restore();
      lm_.lambda = lm_.lambda_vee * lm_.lambda;
      lm_.lambda_vee *= S(opt_.vee_factor);
      lm_.prev_all = lm_.prev_valid = 0;  // the reference leaves it_summary.cost zeroed here
      ++lm_.it;
      if (lm_.lambda > max_lambda) lm_.terminated = true;
      return finish(true);
    }
    rba_residual_info ri2{};
    if (!applied) {
      backup();
      apply(nullptr, &l_diff_d, true);
      compute_error_enqueue(pinned_doubles(kPinCe1));
      stamp_close();
      sync();
      flush_timers();
    }
    if (lm_async_) apply_outcome(&l_diff_d);
    compute_error_parse(pinned_doubles(kPinCe1), &ri2);
    S l_diff = S(l_diff_d);
    row.cost = ri2.all_error;
    row.cost_valid = ri2.valid_error;
    row.num_obs = int(ri2.all_num_obs);
    row.num_obs_valid = int(ri2.valid_num_obs);
    row.residual_sum = ri2.all_residual_sum;
    row.residual_sum_valid = ri2.valid_residual_sum;
    row.l_diff = l_diff;
    if (!std::isfinite(l_diff) || !ri2.is_numerically_valid) {
      row.step_is_valid = row.step_is_successful = 0;
    } else {
      S f_diff;
      if (opt_.optimized_cost == 0)
        f_diff = S(ri.all_error - ri2.all_error);
      else if (opt_.optimized_cost == 1)
        f_diff = S(ri.valid_error - ri2.valid_error);
      else
        f_diff = S(ri.valid_error / std::max(1, ri.valid_num_obs) -
                   ri2.valid_error / std::max(1, ri2.valid_num_obs));
      if (opt_.optimized_cost == 2) l_diff /= S(ri.valid_num_obs);
      const S step_quality = f_diff / l_diff;
      row.relative_decrease = step_quality;
      row.step_is_valid = l_diff > 0;
      row.step_is_successful = row.step_is_valid && step_quality > S(opt_.min_relative_decrease);
    }
    if (row.step_is_successful) {
      lm_.lambda *= S(std::max(1.0 / 3, 1 - std::pow(2 * row.relative_decrease - 1, 3)));
      lm_.lambda = std::max(min_lambda, lm_.lambda);
      lm_.lambda_vee = S(opt_.initial_vee);
      ++lm_.it;
      double cost, change;
      if (opt_.optimized_cost == 0) {
        cost = ri2.all_error;
        change = std::abs(lm_.prev_all - ri2.all_error);
      } else {
        cost = ri2.valid_error;
        change = std::abs(lm_.prev_valid - ri2.valid_error);
      }
      lm_.prev_all = ri2.all_error;
      lm_.prev_valid = ri2.valid_error;
      if (change <= opt_.function_tolerance * cost) {
        lm_.terminated = true;
        lm_.termination = 1;
      }
      lm_.need_linearize = true;
      lm_.ri = ri2;
      lm_.ri_is_current = true;
      return finish(true);
    }
    lm_.lambda = lm_.lambda_vee * lm_.lambda;
    lm_.lambda_vee *= S(opt_.vee_factor);
    lm_.prev_all = ri2.all_error;
    lm_.prev_valid = ri2.valid_error;
    restore();
    ++lm_.it;
    if (lm_.lambda > max_lambda) lm_.terminated = true;
    return finish(true);
  }

  int lm_termination() const override { return lm_.termination; }

  int optimize_lm(rba_lm_iteration* log, int max_rows, int* n_rows_out, int* term_out) override {
    lm_begin();
    int n_rows = 0;
    for (;;) {
      rba_lm_iteration row{};
      const int more = lm_step(&row);
      const bool produced = !(lm_.termination == -1 && row.cg_iterations == 0 && row.cost == 0) &&
                            row.iteration <= opt_.max_num_iterations;
      if (produced) {
        if (n_rows < max_rows) log[n_rows] = row;
        ++n_rows;
      }
      if (!more) break;
    }
    sync();
    *n_rows_out = n_rows;
    *term_out = lm_.termination;
    return RBA_OK;
  }

  // ---- misc -----------------------------------------------------------------------------
  // PMC calibration helper: stream the block storage once; returns bytes read
  int64_t debug_read_A(int vec) override {
    use_device();
    const size_t n = d_JpS_.size() * sizeof(S) / sizeof(float);
    const float* src = reinterpret_cast<const float*>(d_JpS_.get());
    float* sink = reinterpret_cast<float*>(d_tmp_.get());
    if (vec == 4)
      hipLaunchKernelGGL((rba::k_calib_read<4>), dim3(8192), dim3(256), 0, stream_, src, n, sink);
    else
      hipLaunchKernelGGL((rba::k_calib_read<1>), dim3(8192), dim3(256), 0, stream_, src, n, sink);
    sync();
    return int64_t(n / (vec == 4 ? 4 : 1) * (vec == 4 ? 4 : 1)) * 4;
  }
  void device_sync() override {
    use_device();
    sync();
  }
  void get_timings(rba_iter_timings* out) override { *out = timings_; }
  void get_substage_timings(rba_substage_timings* out) override { *out = sub_; }
  void get_jl_col_scale(void* out) override {
    use_device();
    std::vector<S> sorted(3 * size_t(n_lms_));
    d_jl_scale_.download(sorted.data(), sorted.size(), stream_);
    sync();
    S* o = static_cast<S*>(out);
    for (int s = 0; s < n_lms_; ++s)
      for (int c = 0; c < 3; ++c) o[3 * size_t(perm_[s]) + c] = sorted[3 * size_t(s) + c];
  }
  void get_pose_scaling(void* out) override {
    use_device();
    ensure_gram();
    d_pose_scaling_.download(static_cast<S*>(out), nvec_, stream_);
    sync();
  }
  void get_landmark_R(int damped, void* R6, void* q3) override {
    if (sc_) throw HipError{"the SCHUR_COMPLEMENT solver has no triangular factors", RBA_ERR_UNSUPPORTED};
    use_device();
    std::vector<S> R(6 * size_t(n_lms_)), q(3 * size_t(n_lms_));
    if (damped) {
      d_Rd_.download(R.data(), R.size(), stream_);
      d_q1trd_.download(q.data(), q.size(), stream_);
      sync();
    } else {
      d_R0_.download(R.data(), R.size(), stream_);
      std::vector<S> vh(8 * size_t(n_obs_));  // (v0, v1, v2, Q^T r) per block row
      std::vector<int64_t> lm_obs(n_lms_ + 1);
      d_Vh_.download(vh.data(), vh.size(), stream_);
      d_lm_obs_.download(lm_obs.data(), lm_obs.size(), stream_);
      sync();
      for (int s = 0; s < n_lms_; ++s)
        for (int c = 0; c < 3; ++c) q[3 * size_t(s) + c] = vh[4 * (2 * lm_obs[s] + c) + 3];
    }
    S* Ro = static_cast<S*>(R6);
    S* qo = static_cast<S*>(q3);
    for (int s = 0; s < n_lms_; ++s) {
      for (int c = 0; c < 6; ++c) Ro[6 * size_t(perm_[s]) + c] = R[6 * size_t(s) + c];
      for (int c = 0; c < 3; ++c) qo[3 * size_t(perm_[s]) + c] = q[3 * size_t(s) + c];
    }
  }
  // |Q2^T r| per landmark (undamped): the rows >= 3 of the fourth entry of the reflector records
  void get_landmark_q2tr_norm(void* out) override {
    if (sc_) throw HipError{"the SCHUR_COMPLEMENT solver has no triangular factors", RBA_ERR_UNSUPPORTED};
    use_device();
    std::vector<S> vh(8 * size_t(n_obs_));
    std::vector<int64_t> lm_obs(n_lms_ + 1);
    d_Vh_.download(vh.data(), vh.size(), stream_);
    d_lm_obs_.download(lm_obs.data(), lm_obs.size(), stream_);
    sync();
    S* o = static_cast<S*>(out);
    for (int s = 0; s < n_lms_; ++s) {
      double ss = 0;
      for (int64_t r = 2 * lm_obs[s] + 3; r < 2 * lm_obs[s + 1]; ++r) ss += double(vh[4 * r + 3]) * double(vh[4 * r + 3]);
      o[perm_[s]] = S(std::sqrt(ss));
    }
  }
  // compulsory HBM bytes per launch group in this layout (include/rootba_hip.h: rba_byte_model): every record read
  // or written once by the kernel group that needs it, per-landmark and camera-sized data once per kernel. The
  // measured traffic (rocprofv3 FETCH_SIZE / WRITE_SIZE, profiles/) can only be larger: gathers fetch whole cache
  // lines, per-landmark records are re-read by the work-items of a landmark. tests/test_byte_model.py holds the
  // model to that bound on the committed counter tables.
  void get_byte_model(rba_byte_model* m) override {
    const int64_t s = sizeof(S), no = n_obs_, nl = n_lms_, nc = n_cams_;
    const int64_t geometry_in = no * (2 * s + 8) + nl * 3 * s + nc * 10 * s;  // observations, their two indices, points, cameras
    m->compute_error = geometry_in;
    if (sc_) {
      m->stage1 = 2 * geometry_in + no * (26 * s) + nl * 12 * s + nc * 9 * s;
      m->stage2 = nl * 21 * s + no * (26 + 63) * s + sc_assemble_bytes_ + nc * 90 * s;
      m->back_substitution = no * (26 * s + 4) + nl * 18 * s;
      m->product_matrix_free = 0;
    } else {
      const int64_t no_untiled = no - (n_tiles_ > 0 ? n_obs_tiled_ : 0);
      // stage 1: geometry writes JpS 18 + Vh 8; the QR pass reads Vh 8 (+ the row map, 4 B per block row) and writes
      // Vh 8 (+ JlS 6 + rS 2 for the untiled landmarks) per observation, R0 6 + tau 3 + LQ 12 + Jl_col_scale 3 per landmark
      m->stage1 = geometry_in + no * ((18 + 8) + (8 + 8)) * s + no_untiled * 8 * s + no * 8 + nl * 24 * s;
      // fused geometry + QR of the tiled landmarks (k_s1_fused_obs): no Vh 8 written and read back, the camera map
      // of the tile (4 B per block row) instead of the two indices of the observation
      if (s1_fused()) m->stage1 -= (no - no_untiled) * 16 * s;
      // the Gram pass on its own (JpS 18 + CSC index in, G 81 + Jp_diag2 9 out): not on one GPU, where it rides on the
      // camera pass of the first stage 2, which reads those rows anyway (already counted there)
      if (comm_ || cb_fn_ || !opt_.staged_execution) m->stage1 += no * (18 * s + 4) + nc * 90 * s;
      // stage 2, landmark side (k_s2_obs): Vh 8 + landmark index in, WA 8 out per observation (+ the eight coefficients
      // of the untiled ones); R0 6 + LQ 12 + the first three reflector rows 12 in, givens 16 + Rd 6 + Q1^T r 3 +
      // damping residual 3 + Z 9 out per landmark. Camera pass: JpS 18 + WA 8 + CSC index in; blocks 81, b 9,
      // Jp_diag2 9, scaling 9 out. (Round 6, split storage of the rows: the landmark side reads the two tail entries of
      // an observation's rows and writes them into WA[6..7]; the camera pass reads the 16 main entries + WA 8 - the
      // same 42 scalars per observation in all.)
      m->stage2 = no * (10 * s + 4) + no * 8 * s + no_untiled * 8 * s + nl * (30 + 37) * s + no * (24 * s + 4) +
                  nc * 108 * s;
      // back-substitution, one pass on the wave tiles (k_bs_tile): JpS 18 + Vh 8 and the tile map (8 B per observation
      // since round 6: Params::OT) per observation = 112 B in float; tau 3 + givens 16 + R0 6 + Rd 6 + Q1^T r 3 + Jl_col_scale 3 in,
      // the point in and out (6) and l_diff (8 B) out per landmark. The two-kernel form of the untiled landmarks adds
      // JlS 6 + rS 2, the eight coefficients and its 5-scalar scratch (write + read) per observation.
      m->back_substitution = no * (26 * s + 8) + no_untiled * (8 + 8 + 10) * s + nl * (43 * s + 8) + nc * 9 * s;
      // implicit-Q product: JpS row 9 + Vh row 4 per block row, the tile map (8 B per observation), tau + Z per landmark
      m->product_matrix_free = no * (26 * s + 8) + nl * 12 * s + nc * 18 * s;
    }
    const int64_t nnz = sc_ ? sc_nnz_ : ex_nnz_;
    const int64_t ms = sc_ ? s : int64_t(sizeof(double));  // the assembled matrix of the square-root solver is double
    m->product_assembled = nnz * (81 * ms + 4) + nc * 18 * s;
    m->pcg_vectors = nc * (81 + 10 * 9) * s;
    // persistent kernel (kernels_pcgp.hpp): per solve the FULL-storage matrix once (2 nnz - n_c blocks of 81 doubles out
    // of the half-storage one), the row state and M^-1; per iteration only the exchanged records - 16 bytes per three
    // entries of z (one per entry in double), the partial sums in their replicas
    m->persistent_solve = pg_ready_ ? (2 * nnz - nc) * 81 * 8 + nc * (81 + 4 * 9) * s : 0;
    m->persistent_iteration = pg_ready_ ? nc * rba::pg_vec_records<S>() * 16 * 2 + int64_t(pg_G_) * 3 * 16 * (1 + rba::kPgReplicas) : 0;
    if (!sc_) {
      // half storage: the transposed parts (9 doubles per block off the diagonal) + their slot index go out with the
      // product and come back in with the vector kernel that consumes q
      // (round 5: BOTH directions of the slots are priced with the product - the vector work of an iteration is then the
      //  same number for an iteration on either operator, which is what `iterations - iterations_resident` multiplies:
      //  with the persistent kernel those are the matrix-free iterations, which have no slots)
      m->product_assembled += int64_t(nnz - nc) * (9 * 8) * 2 + int64_t(nnz) * 4;
    }
    // assembly, COMPULSORY bytes: the pair list (8 B per pair), every 32-scalar record once, the blocks out. The
    // gather itself requests two records per pair (256 B in float); what L2 does not keep of that is re-read traffic
    // and shows up as measured / model > 1 (profiles/r3_pmc_stage_traffic.csv), not as algorithmic bytes.
    m->assembly = sc_ ? sc_assemble_bytes_ : ex_pairs_ * 8 + int64_t(n_obs_) * 32 * ms + nnz * 81 * ms;
    if (!sc_)  // + the column pass that materialises the records of damped top rows: JpS 18 + Vh 8 in, 32 out
      m->assembly += int64_t(n_obs_) * ((18 + 8) * s + 32 * ms);
    if (!sc_ && kA64)  // float solver (kernels_a64.hpp): + the landmark pass (Vh 8 in), the factor A (4 doubles) out and
      m->assembly += int64_t(n_obs_) * (8 * s + 4 * ms + 18 * s + 4 * ms + 4);  // back in with JpS 18 + the CSC index

  }
  void get_pcg_counters(rba_pcg_counters* out) override { *out = pcg_counters_; }
  void get_reduced_matrix_info(rba_reduced_matrix_info* out) override {
    const int64_t nc = n_cams_;
    const bool have = sc_ ? sc_nnz_ > 0 : ex_ready_;
    out->blocks_stored = have ? (sc_ ? sc_nnz_ : ex_nnz_) : 0;
    out->blocks_full = have ? (sc_ ? int64_t(sc_nnz_) : 2 * int64_t(ex_nnz_) - nc) : 0;
    out->density = have && nc > 0 ? double(out->blocks_full) / (double(nc) * double(nc)) : 0.0;
    out->bytes_stored = out->blocks_stored * 81 * int64_t(sc_ ? sizeof(S) : sizeof(double));
    out->resident_in_registers = pcg_persistent_possible() ? 1 : 0;
    out->persistent_workgroups = pg_ready_ ? pg_G_ : 0;
  }
  void get_problem_stats(int64_t* storage, int64_t* hx_bytes, int64_t* hx_flops) override {
    *storage = storage_bytes_;
    *hx_bytes = sc_ ? hx_bytes_ : hx_implicit_bytes_;
    *hx_flops = hx_flops_;
  }

 private:
  int ce_blocks_ = 0;  // compute_error_blocks()
  static constexpr int kReduceBlocks = 2048;  // = the wave slots of the chip at 256 threads x 8 waves per SIMD
  static constexpr size_t kSmallLdsBudget = 16 * 1024;
  static constexpr int kSmallBatchesPerBlock = 8;  // LDS batches walked by one workgroup  // bytes of A per small-landmark batch
  static constexpr int kMaxHxEvents = 1024;

  void use_device() { HIP_CHECK(hipSetDevice(device_)); }
  // result slots in the pinned page h_pinned_ (the PCG state copy lives at offset 0)
  static constexpr size_t kPinCe0 = 1024, kPinCe1 = 1024 + 64, kPinLdiff = 1024 + 128, kPinFailLin = 1024 + 136,
                          kPinFailApply = 1024 + 140, kPinCheck = 1024 + 192, kPinInc = 1024 + 256, kPinEnd = 1024 + 320;
  double* pinned_doubles(size_t off) { return reinterpret_cast<double*>(h_pinned_ + off); }
  int* pinned_int(size_t off) { return reinterpret_cast<int*>(h_pinned_ + off); }
  void sync() {
    join_side();
    HIP_CHECK(hipStreamSynchronize(stream_));
  }
  // Host side of the run-ahead throttles: spin on pinned words the device publishes its progress to, with a pause per
  // turn (the sibling hardware thread - possibly the one that feeds another rank's stream - gets the core). Every
  // 16384 turns the stream is queried: an idle stream ends the wait (everything queued has run: the caller queues
  // more), an error is reported at once instead of at the next synchronisation.
  template <class Pred>
  void spin_while(Pred&& pred) {
    long spins = 0;
    while (pred()) {
      cpu_relax();
      if ((++spins & 0x3fff) == 0) {
        const hipError_t e = hipStreamQuery(stream_);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) HIP_CHECK(e);
      }
    }
  }
  // Stage timers: HIP-event pairs on the solver stream. A single C-ABI call waits for its own pair; inside
  // rba_lm_step (lm_async_) the pairs are only RECORDED and read after the iteration's last synchronisation, so the
  // host never stalls the queue just to read a clock (round 2: ~10 synchronisations per LM iteration, 0.14 ms of gaps).
  struct PendingTimer {
    hipEvent_t e0, e1;
    double* field;
    bool accumulate;
  };
  hipEvent_t timer_event() {
    if (timer_pool_used_ == timer_pool_.size()) {
      hipEvent_t e = nullptr;
      HIP_CHECK(hipEventCreate(&e));
      timer_pool_.push_back(e);
    }
    return timer_pool_[timer_pool_used_++];
  }
  // Inside rba_lm_step (lm_async_) the stage boundaries are DEVICE stamps instead of events: the first kernel of a stage
  // writes the chip-wide 100 MHz clock into a pinned slot when it starts (kernels.hpp: stage_stamp) - the end of a stage
  // is the start of the next one's first kernel on the same stream, or the end of its own last kernel where that is a
  // single workgroup (k_reduce_rows) - and the slots are read after the iteration's synchronisation. (A HIP event is a
  // marker packet of ~4 us on the queue: the twelve of an LM iteration cost 50 us - 3.5 % of a venice iteration, 15 % of
  // a trafalgar-257 iteration, measured with RBA_STAGE_TIMERS=0.) RBA_STAGE_TIMERS=2: events everywhere, as before.
  static constexpr int kMaxStamps = 256;
  struct PendingStamp {
    int s0, s1;
    double* field;
    bool accumulate;
  };
  bool stamps_on() const { return lm_async_ && env_.stage_timers == 1 && h_stamps_; }
  int stamp_slot() {
    if (stamp_n_ >= kMaxStamps) return -1;
    h_stamps_[stamp_n_] = 0;
    return stamp_n_++;
  }
  // the pending stage boundary, for the kernel that is launched next on the solver stream (nullptr: none pending)
  unsigned long long* stamp() {
    if (!stamps_on() || stamp_pending_ < 0) return nullptr;
    unsigned long long* p = h_stamps_ + stamp_pending_;
    stamp_pending_ = -1;
    return p;
  }
  rba::Params<S> prm_stamped() {
    rba::Params<S> p = prm_;
    p.stamp = stamp();
    return p;
  }
  // a boundary that no kernel has taken before the stream is synchronised
  void stamp_close() {
    if (unsigned long long* p = stamp()) hipLaunchKernelGGL(rba::k_stamp, dim3(1), dim3(1), 0, stream_, p);
  }
  void time_begin(hipStream_t st = nullptr) {
    if (!env_.stage_timers) return;
    if (stamps_on()) {
      if (st && st != stream_) {  // (side stream: a pair of its own, see time_end)
        stamp_side_slot_ = stamp_slot();
        stamp_side_ = stamp_side_slot_ >= 0 ? h_stamps_ + stamp_side_slot_ : nullptr;
        return;
      }
      if (stamp_pending_ < 0) stamp_pending_ = stamp_slot();
      stamp_begin_ = stamp_pending_;
      return;
    }
    timer_t0_ = timer_event();
    HIP_CHECK(hipEventRecord(timer_t0_, st ? st : stream_));
  }
  // `end_slot` >= 0: the stage's last kernel has stamped its own end there (stamp_end_slot())
  void time_end(double* field, bool accumulate = false, hipStream_t st = nullptr, int end_slot = -1) {
    if (!env_.stage_timers) return;
    if (stamps_on()) {
      int e = end_slot;
      if (st && st != stream_) {  // side stream: begin stamped by the stage's first kernel, end by its last
        if (stamp_side_slot_ >= 0 && e >= 0) pending_stamps_.push_back(PendingStamp{stamp_side_slot_, e, field, accumulate});
        stamp_side_slot_ = -1;
        return;
      }
      if (e < 0) {
        if (stamp_pending_ >= 0 && stamp_pending_ == stamp_begin_) {
          e = stamp_begin_;  // no kernel took the begin boundary: an empty stage
        } else {
          e = stamp_slot();
          stamp_pending_ = e;
        }
      }
      if (stamp_begin_ >= 0 && e >= 0) pending_stamps_.push_back(PendingStamp{stamp_begin_, e, field, accumulate});
      return;
    }
    hipEvent_t e1 = timer_event();
    HIP_CHECK(hipEventRecord(e1, st ? st : stream_));
    pending_timers_.push_back(PendingTimer{timer_t0_, e1, field, accumulate});
    if (!lm_async_) {
      HIP_CHECK(hipEventSynchronize(e1));
      flush_timers();
    }
  }
  // (all recorded pairs have completed: the caller synchronised the stream or the last pair's end event)
  void flush_timers() {
    for (const PendingTimer& t : pending_timers_) {
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, t.e0, t.e1));
      *t.field = (t.accumulate ? *t.field : 0.0) + double(ms) * 1e-3;
    }
    pending_timers_.clear();
    timer_pool_used_ = 0;
    for (const PendingStamp& t : pending_stamps_) {
      const unsigned long long a = h_stamps_[t.s0], b = h_stamps_[t.s1];
      const double sec = (a && b && b >= a) ? double(b - a) / stamp_hz_ : 0.0;  // (0: a boundary nobody stamped)
      *t.field = (t.accumulate ? *t.field : 0.0) + sec;
    }
    pending_stamps_.clear();
    stamp_n_ = 0;
    stamp_pending_ = -1;
    stamp_begin_ = -1;
    stamp_side_slot_ = -1;
  }
  void reset_timings() {
    timings_ = rba_iter_timings{};
    sub_ = rba_substage_timings{};
  }
  // Sub-stage timers of the reference's UNSTAGED execution (linearizor_qr.cpp:94-112, 166-187): with
  // staged_execution = 0 the kernel groups of a stage are separated by HIP events (a marker packet each) and the
  // elapsed times are filed under the reference's IterationSummary fields; the kernels are the same either way.
  bool sub_timing() const { return !opt_.staged_execution; }
  // one rank: sums and failure words need no all-reduce, the kernels that produce them write the pinned host page
  bool results_go_direct() const { return !comm_ && !cb_fn_; }
  void sub_begin() {
    sub_n_ = 0;
    if (sub_timing()) sub_mark(nullptr);
  }
  void sub_mark(double* field) {
    if (!sub_timing()) return;
    if (sub_n_ >= int(sub_events_.size())) {
      hipEvent_t e = nullptr;
      HIP_CHECK(hipEventCreate(&e));
      sub_events_.push_back(e);
      sub_fields_.push_back(nullptr);
    }
    HIP_CHECK(hipEventRecord(sub_events_[sub_n_], stream_));
    sub_fields_[sub_n_] = field;
    ++sub_n_;
  }
  // after the stage's time_end() (which synchronised): interval i-1 -> i is added to the field of mark i
  void sub_collect() {
    for (int i = 1; i < sub_n_; ++i) {
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, sub_events_[i - 1], sub_events_[i]));
      if (sub_fields_[i]) *sub_fields_[i] += double(ms) * 1e-3;
    }
    sub_n_ = 0;
  }

  template <class F>
  void for_each_class(F&& f) {
    if (cls_end_[0] > cls_begin_[0]) f(std::integral_constant<int, 1>{}, cls_begin_[0], cls_end_[0]);
    if (cls_end_[1] > cls_begin_[1]) f(std::integral_constant<int, 2>{}, cls_begin_[1], cls_end_[1]);
    if (cls_end_[2] > cls_begin_[2]) f(std::integral_constant<int, 4>{}, cls_begin_[2], cls_end_[2]);
    if (cls_end_[3] > cls_begin_[3]) f(std::integral_constant<int, 8>{}, cls_begin_[3], cls_end_[3]);
    if (cls_end_[4] > cls_begin_[4]) f(std::integral_constant<int, 16>{}, cls_begin_[4], cls_end_[4]);
  }

  // Debug / test environment, read ONCE per handle in construct() (none is needed in production; include/rootba_hip.h
  // lists them). Everything that selects between kernels of equal results is a test hook here, not API.
  struct DebugEnv {
    int verbose = 0;                   // RBA_VERBOSE=1: break-even / budget messages on stderr
    int explicit_after = INT_MIN;      // RBA_EXPLICIT_AFTER=n: overrides rba_options.explicit_after
    double pair_budget_gb = 24.0;      // RBA_EX_PAIR_BUDGET_GB=x: budget of the pair lists of the assembled matrix
    int force_explicit_fallback = 0;   // RBA_FORCE_EXPLICIT_FALLBACK=1: treat every solve on the assembled matrix as broken down
    int hx_lds = 1;                    // RBA_HX_LDS=0 never / 1 automatic / 2 whenever possible: LDS-private products
    int hx_win = 0;                    // RBA_HX_WIN=n: cap of the camera window of the LDS-private products
    int hx_timing_stride = -1;         // RBA_HX_TIMING_STRIDE=n: HIP events around every n-th matrix-free product
    int sort_by_camera = -1;           // RBA_SORT_BY_CAMERA=0/1: override the automatic choice
    int verify_assembled = 0;          // RBA_VERIFY_ASSEMBLED=1: one-product check of assembled-operator solves (diagnostic)
    double verify_tolerance = 0.25;    // RBA_VERIFY_TOLERANCE=x: relative agreement of the Q model asked of them
    int spmv_stream = 1;               // RBA_SPMV_STREAM: 0 = one wavefront per item always, 1 = streaming SpMV for matrices of
                                       // >= 4 items per resident wavefront, 2 = always (tests)
    int spmv_stream_waves_per_cu = 0;  // RBA_SPMV_STREAM_WAVES: wavefronts per compute unit of the streaming SpMV (0 = 4; -1: two in all - tests)
    int spmv_stream_buffers = 1;       // RBA_SPMV_STREAM_BUFFERS=2: the streaming product with two chunks in flight per wavefront and four
                                       // wavefronts per compute unit (default: one chunk, seven wavefronts; k_pcgs_spmv_stream1)
    int series_f32 = 1;                // RBA_SERIES_F32=0: the terms of the power-series preconditioner through the double matrix
    int pcg_persistent = 1;            // RBA_PCG_PERSISTENT=0: PCG on the assembled matrix always in two launches per
                                       // iteration (kernels_pcg.hpp; the test of the two forms)
    int pcg_split = -1;                // RBA_PCG_SPLIT=0/1: never / always split the products on the assembled matrix
                                       // over the ranks (default: where the estimate says it pays)
    int hx_wide_inside = 1;            // RBA_HX_WIDE_INSIDE=0: landmarks with 32 < k <= 64 in a kernel of their own
    int stage_timers = 1;              // RBA_STAGE_TIMERS=0: no HIP events around the stages (rba_iter_timings stays zero)
    int test_fail_rank = -1;           // RBA_TEST_FAIL_RANK=r: rank r of a sharded run throws at its next linearisation (test of the abort path)
    int pcgp_test_give_up = 0;         // RBA_PCGP_TEST_GIVE_UP=1: test hook of the collective fallback of a sharded run
    int deterministic = 0;             // RBA_DETERMINISTIC=1: matrix-free products summed camera-major in a fixed order
                                       // (no floating-point atomics anywhere: runs repeat bit by bit; ~2 x per product)
    int s1_fused = 1;                  // RBA_S1_FUSED=0: geometry and QR of the wave-tile landmarks as two kernels
    int half_lower_max = rba::kHalfLowerMax;  // RBA_HALF_LOWER_MAX=n: received slots above which a row of the assembled
                                              // matrix has them summed by a wavefront of its own (k_pcgs_reduce_slots)
  };
  DebugEnv env_;
  void read_debug_env() {
    auto geti = [](const char* name, int dflt) {
      const char* ev = std::getenv(name);
      return ev ? std::atoi(ev) : dflt;
    };
    env_.verbose = geti("RBA_VERBOSE", 0);
    env_.explicit_after = geti("RBA_EXPLICIT_AFTER", INT_MIN);
    if (const char* ev = std::getenv("RBA_EX_PAIR_BUDGET_GB")) env_.pair_budget_gb = std::atof(ev);
    env_.force_explicit_fallback = geti("RBA_FORCE_EXPLICIT_FALLBACK", 0);
    env_.hx_lds = geti("RBA_HX_LDS", 1);
    env_.hx_win = geti("RBA_HX_WIN", 0);
    env_.hx_timing_stride = geti("RBA_HX_TIMING_STRIDE", -1);
    env_.sort_by_camera = geti("RBA_SORT_BY_CAMERA", -1);
    env_.verify_assembled = geti("RBA_VERIFY_ASSEMBLED", 0);
    if (const char* ev = std::getenv("RBA_VERIFY_TOLERANCE")) env_.verify_tolerance = std::atof(ev);
    env_.half_lower_max = geti("RBA_HALF_LOWER_MAX", rba::kHalfLowerMax);
    env_.s1_fused = geti("RBA_S1_FUSED", 1);
    env_.hx_wide_inside = geti("RBA_HX_WIDE_INSIDE", 1);
    env_.deterministic = geti("RBA_DETERMINISTIC", 0);
    env_.pcgp_test_give_up = geti("RBA_PCGP_TEST_GIVE_UP", 0);
    env_.test_fail_rank = geti("RBA_TEST_FAIL_RANK", -1);
    env_.stage_timers = geti("RBA_STAGE_TIMERS", 1);
    env_.pcg_split = geti("RBA_PCG_SPLIT", -1);
    env_.pcg_persistent = geti("RBA_PCG_PERSISTENT", 1);
    env_.series_f32 = geti("RBA_SERIES_F32", 1);
    env_.spmv_stream = geti("RBA_SPMV_STREAM", 1);
    env_.spmv_stream_waves_per_cu = geti("RBA_SPMV_STREAM_WAVES", 0);
    env_.spmv_stream_buffers = geti("RBA_SPMV_STREAM_BUFFERS", 1);
    if (env_.hx_timing_stride >= 0) hx_timing_stride_ = env_.hx_timing_stride;
  }

  int device_;
  int n_cams_, n_lms_;
  int64_t n_obs_ = 0;
  int nvec_ = 0;
  rba_options opt_;
  hipStream_t stream_ = nullptr;
  hipStream_t side_stream_ = nullptr;  // cost evaluation beside the linearisation (compute_error_enqueue)
  hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
  bool side_pending_ = false;
  DevBuf<double> d_partials_side_;
  DevBuf<double> d_endred_;  // rba_lm_step of a sharded run: everything the end of an iteration all-reduces, in one piece (rba::kEndRed)
  std::vector<int> perm_;
  int cls_begin_[kNumClasses], cls_end_[kNumClasses];
  int big_begin_ = 0, n_big_ = 0, big_kmax_ = 0;
  static constexpr int kNumImplicit = 7;
  int imp_begin_[kNumImplicit], imp_end_[kNumImplicit];
  int64_t hx_bytes_ = 0, hx_flops_ = 0, storage_bytes_ = 0, hx_implicit_bytes_ = 0;
  int64_t ex_pairs_ = 0, ex_pair_bytes_ = 0;
  rba::Params<S> prm_{};
  S pose_damping_ = S(0);
  bool landmark_damping_valid_ = false;
  rba_iter_timings timings_{};
  // device buffers
  DevBuf<int> d_lm_k_, d_obs_cam_, d_obs_lm_, d_fail_;
  DevBuf<int64_t> d_lm_obs_, d_cam_off_;
  DevBuf<int> d_cam_obs_;
  DevBuf<S> d_JpS_, d_Vh_, d_tauH_, d_Zd_, d_LQ_, d_JlS_, d_rS_, d_bsO_, d_givens_;
  DevBuf<int2> d_OT_;  // tile map (kernels.hpp: Params::OT)
  int64_t n_obs_small_ = 0, n_obs_tiled_ = 0;
  DevBuf<S> d_big_scratch_;
  DevBuf<S> d_pg_xsave_;                  // sharded runs: x and the PCG state at the entry of a persistent solve
  DevBuf<rba::CgState> d_pg_stsave_;
  DevBuf<S> d_hx_u_, d_e0_w_;  // RBA_DETERMINISTIC=1: [2 n_obs] row entries of P J x, [3 n_lms] landmark sums of E0 v
  DevBuf<int64_t> d_big_off_;
  int imp_tile_begin_[5] = {0, 0, 0, 0, 0}, imp_tiles_[5] = {0, 0, 0, 0, 0}, n_tiles_ = 0;
  DevBuf<S> d_obs_xy_, d_cams_, d_lms_, d_cams_bak_, d_lms_bak_;
  // RBA_MIXED (S = float): double master state, double observations, float landmark increments
  bool mixed_ = false;
  DevBuf<double> d_obs_xy64_, d_cams64_, d_lms64_, d_cams64_bak_, d_lms64_bak_;
  DevBuf<S> d_lm_inc_;
  rba::Params<double> prm64_{};
  DevBuf<S> d_topd_, d_R0_, d_Rd_, d_q1trd_, d_damp_r_, d_jl_scale_;
  DevBuf<S> d_jp_diag2_, d_pose_scaling_, d_mid_, d_bb_, d_inv_;
  DevBuf<S> d_x_, d_r_, d_p_, d_z_, d_q_, d_tmp_, d_inc_, d_vin_, d_pw_t_, d_pw_e_;
  DevBuf<double> d_lm_ldiff_, d_partials_, d_pcg_partials_;
  DevBuf<rba::CgState> d_cg_;
  char* h_pinned_ = nullptr;
  S* h_vec_stage_ = nullptr;  // pinned staging of the camera-sized vectors that cross the C ABI per LM iteration (the
                              // increment out of rba_solve, into rba_apply): a copy to / from pageable memory is staged
                              // and synchronised by the runtime
  hipEvent_t timer_t0_ = nullptr;
  std::vector<hipEvent_t> timer_pool_;
  size_t timer_pool_used_ = 0;
  std::vector<PendingTimer> pending_timers_;
  unsigned long long* h_stamps_ = nullptr;  // pinned: stage boundaries of rba_lm_step (time_begin / time_end)
  unsigned long long* stamp_side_ = nullptr;
  int stamp_n_ = 0, stamp_pending_ = -1, stamp_begin_ = -1, stamp_side_slot_ = -1;
  double stamp_hz_ = 1e8;
  std::vector<PendingStamp> pending_stamps_;
  // rba_lm_step (staged execution): launches folded into their neighbours (round 6)
  bool lm_fuse_ = false;              // the step is running in that mode
  bool lin_flag_deferred_ = false;    // the failure word of the linearisation has not been published yet
  bool inc_prescaled_ = false;        // d_xs_ holds D inc (written by the end of the solve)
  bool cams_backup_in_update_ = false;  // k_update_cameras leaves the cameras it replaces in the backup buffer
  bool lm_merge_end_ = false;         // sharded run: l_diff, failure bits and the trial cost sums in ONE all-reduce
  bool lm_end_merged_ = false;        // ... and that block has been queued for the host
  bool in_lm_step_ = false;  // rba_lm_step is running: apply / restore are ITS calls (the cached cost stays valid)
  bool lm_async_ = false;  // inside rba_lm_step: results and timers are collected at the iteration's own sync points
  std::vector<hipEvent_t> hx_events_;
  std::vector<int> hx_event_call_;  // which H*x call of the solve each event pair brackets
  int hx_event_count_ = 0, hx_calls_ = 0;
  bool operand_prescaled_ = false;
  DevBuf<S> d_WA_;             // stage-2 records of the camera-major pass (kernels_cam.hpp)
  bool gram_pending_ = false;  // linearised without the Gram pass: the first stage 2's camera pass does it
  bool topd_valid_ = false;    // the 27 + 9 records exist for the current damping (assembly / E0 products only)
  DevBuf<S> d_W8_, d_xs_;
  rba_substage_timings sub_{};
  std::vector<hipEvent_t> sub_events_;
  std::vector<double*> sub_fields_;
  int sub_n_ = 0;
  bool hx_lds_attr_set_ = false;
  DevBuf<rba::HxChunk> d_hx_chunks_;
  int n_hx_chunks_ = 0, hx_win_ = 0;
  double hx_coverage_ = 1.0;  // share of the tiled block rows whose camera lies inside its workgroup's LDS window
  static constexpr size_t kHxLdsMaxBytes = 152 * 1024;  // of the CU's 160 KB
  int n_cus_ = 256;
  bool solve_defer_ = false;         // rba_lm_step: solve() leaves its host-side collection to solve_collect()
  bool pcg_state_pending_ = false;   // the PCG ended without a host copy of its state (k_finish_increment carries it)
  int pcg_it_first_assembled_ = 1;
  rba_cg_summary solve_cg_{0, 0};
  // explicit reduced matrix of the square-root solver (adaptive, see pcg())
  int explicit_after_ = 0;  // matrix-free products before a solve switches to S x; 0 = never
  bool ex_ready_ = false, ex_valid_ = false, ex_active_ = false;
  bool explicit_auto_ = false, asm_measured_ = false, asm_pending_ = false;
  bool early_switch_pays_ = true;        // (until the break-even has been measured)
  bool pcg_last_was_persistent_ = false; // the last solve on the assembled matrix ran as the persistent kernel
  int64_t last_mf_products_ = 0, last_as_products_ = 0;  // products of the last solve on either operator
  double measured_t_as_ms_ = 0;
  hipEvent_t ev_asm0_ = nullptr, ev_asm1_ = nullptr;
  DevBuf<int> d_scratch_int_;
  int ex_nnz_ = 0, ex_n_upper_ = 0;
  std::vector<std::vector<int>> ex_nb_;  // co-observing cameras per camera (structure of the reduced matrix)
  rba::ScParams<S> exp_{};
  std::vector<int64_t> h_lm_obs_;  // host copies of the (sorted) topology for the pair lists
  std::vector<int> h_obs_cam_;
  DevBuf<int> d_ex_rowptr_, d_ex_cols_, d_ex_diag_, d_ex_upper_, d_ex_mirror_, d_ex_pair_oi_, d_ex_pair_oj_;
  DevBuf<int64_t> d_ex_pair_ptr_;
  DevBuf<double> d_ex_vals_;  // always double (assemble_values), half storage (kernels_pcg.hpp)
  DevBuf<float> d_ex_vals32_;  // its float copy for the terms of the power-series preconditioner (series_f32())
  DevBuf<double> d_tpart_;    // [9 n_slots] transposed contributions of the blocks other rows own, per product
  DevBuf<int> d_low_ptr_, d_tdst_;
  DevBuf<rba::HeavyRow> d_heavy_;  // rows whose received slots are summed by a wavefront of their own
  int n_heavy_ = 0;
  bool split_ = false, split_partial_ = false;  // products on the assembled matrix split over the ranks (decide_product_split)
  std::vector<size_t> split_slot_bounds_;       // ... and the ranks' ranges of the matrix, in scalars (reduce_ranges)
  int split_item0_ = 0, split_item1_ = 0;
  // float solver: double re-derivation of the factors for the assembled matrix (kernels_a64.hpp)
  static constexpr bool kA64 = std::is_same<S, float>::value;
  DevBuf<double> d_a64_lq_, d_a64_A_, d_a64_rec_;
  rba::A64Params a64_{};
  bool a64_lm_valid_ = false;  // per linearisation point
  DevBuf<double> d_gram64_;    // [81 n_c] D Hpp D in double (JACOBI / power-series preconditioners)
  bool gram64_valid_ = false;  // per linearisation point
  // fused PCG on the assembled matrix (kernels_pcg.hpp)
  DevBuf<rba::SpmvItem> d_items_;
  DevBuf<int> d_item_ptr_;
  DevBuf<S> d_qpart_, d_qmain_, d_p2_;  // extra-item partials, q, second direction buffer
  DevBuf<double> d_pcgs_pq_;
  int n_items_ = 0;
  int* h_progress_ = nullptr;  // pinned: [0] iteration started on the device, [1] done
  bool pcg_used_explicit_ = false, pcg_indefinite_ = false, explicit_off_for_solve_ = false;
  rba_pcg_counters pcg_counters_{};
  hipGraphExec_t pcg_graph_exec_[2] = {nullptr, nullptr};
  // persistent PCG with the matrix in the register files (kernels_pcgp.hpp)
  bool pg_ready_ = false, pg_broken_ = false, pcg_last_persistent_ = false;
  int pg_G_ = 0;
  unsigned pg_epoch_ = 1;  // tag base of the next solve: tags never repeat between launches
  DevBuf<rba::PgWorkgroup> d_pg_wg_;
  DevBuf<int> d_pg_lane_src_, d_pg_stage_col_, d_pg_row_info_;
  DevBuf<unsigned short> d_pg_lane_col_;
  DevBuf<rba::pg_rec> d_pg_zg_, d_pg_xg_, d_pg_tg_, d_pg_part_;  // 16-byte records of the exchanged vectors and partial sums
  // explicit Schur-complement backend (solver_type = 1)
  bool sc_ = false;
  int sc_nnz_ = 0;
  rba::ScParams<S> scp_{};
  DevBuf<S> d_sc_JlS_, d_sc_rS_, d_sc_M_, d_sc_v_, d_sc_Hinv_, d_sc_hb_, d_sc_W_, d_sc_T_, d_sc_bO_, d_sc_vals_;
  DevBuf<int> d_sc_rowptr_, d_sc_cols_, d_sc_diag_, d_sc_pair_oi_, d_sc_pair_oj_, d_sc_upper_, d_sc_mirror_;
  int sc_n_upper_ = 0;
  DevBuf<int64_t> d_sc_pair_ptr_;
  int64_t sc_assemble_bytes_ = 0;
  int hx_timing_stride_ = 8;  // HIP events around every n-th H*x (rba_iter_timings.hx_time); 0 = off
  // LM state machine
  struct LmState {
    bool active = false, terminated = false, need_linearize = true;
    bool ri_is_current = false;  // `ri` belongs to the current state (set after an accepted step)
    int it = 0, termination = 0;
    S lambda = S(0), lambda_vee = S(0);
    double prev_all = 0, prev_valid = 0;
    rba_residual_info ri{};
  };
  LmState lm_;
  // multi-GPU
  void* comm_ = nullptr;
  std::atomic<bool> comm_aborted_{false};
  int rank_ = 0, nranks_ = 1;
  rba_allreduce_fn cb_fn_ = nullptr;
  void* cb_ctx_ = nullptr;
  std::vector<char> cb_stage_;
  static constexpr int kCommEvents = 64;
  std::vector<hipEvent_t> comm_events_;
  int64_t comm_ev_head_ = 0, comm_ev_tail_ = 0, comm_timed_ = 0, comm_untimed_ = 0;
  int64_t comm_calls_ = 0, comm_bytes_ = 0;
  double comm_seconds_ = 0;
};

template <class F>
int guarded(F&& f) {
  try {
    return f();
  } catch (const HipError& e) {
    g_last_error = e.msg;
    return e.code;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return RBA_ERR_INVALID_ARGUMENT;
  }
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
namespace {

// what rba_create checks before any HIP resource exists (the options the reference CHECKs, the arguments)
int validate_create(const char* who, int32_t n_cams, int32_t n_lms, const int64_t* lm_obs_offsets, const int32_t* obs_cam_idx,
                    const void* obs_xy, const rba_options* options, rba_handle* out) {
  if (!out || !lm_obs_offsets || !obs_cam_idx || !obs_xy || !options || n_cams <= 0 || n_lms <= 0) {
    g_last_error = std::string(who) + ": invalid argument";
    return RBA_ERR_INVALID_ARGUMENT;
  }
  if (options->preconditioner_type < 0 || options->preconditioner_type > 2) {
    // the reference LOG(FATAL)s for anything but JACOBI / SCHUR_JACOBI in the QR
    // solver (linearizor_qr.cpp:208-240); POWER_SCHUR_COMPLEMENT (2) is the new
    // combination of BASELINE.json config 5
    g_last_error = "preconditioner_type must be JACOBI (0), SCHUR_JACOBI (1) or POWER_SCHUR_COMPLEMENT (2)";
    return RBA_ERR_UNSUPPORTED;
  }
  {
    const rba_options& o = *options;
    const char* bad = nullptr;
    if (o.max_cg_it < 1) bad = "max_cg_it (max_linear_solver_iterations) must be >= 1";
    else if (o.min_cg_it < 0 || o.min_cg_it > o.max_cg_it) bad = "0 <= min_cg_it <= max_cg_it required";
    else if (!(o.eta >= 0.0)) bad = "eta must be >= 0";
    else if (o.power_order < 0) bad = "power_order must be >= 0";
    else if (!(o.min_trust_region_radius > 0.0) || !(o.initial_trust_region_radius >= o.min_trust_region_radius) ||
             !(o.max_trust_region_radius >= o.initial_trust_region_radius))
      bad = "0 < min_trust_region_radius <= initial_trust_region_radius <= max_trust_region_radius required";
    else if (o.max_num_iterations < 0) bad = "max_num_iterations must be >= 0";
    else if (!(o.initial_vee > 0.0) || !(o.vee_factor > 0.0)) bad = "initial_vee and vee_factor must be > 0";
    else if (o.robust_norm < 0 || o.robust_norm > 1) bad = "robust_norm must be NONE (0) or HUBER (1)";
    else if (o.robust_norm == 1 && !(o.huber_parameter > 0.0)) bad = "huber_parameter must be > 0";
    else if (o.optimized_cost < 0 || o.optimized_cost > 2) bad = "optimized_cost must be 0, 1 or 2";
    else if (o.solver_type < 0 || o.solver_type > 1) bad = "solver_type must be SQUARE_ROOT (0) or SCHUR_COMPLEMENT (1)";
    else if (!(o.jacobi_scaling_eps >= 0.0)) bad = "jacobi_scaling_eps must be >= 0";
    if (bad) {
      g_last_error = std::string(who) + ": " + bad;
      return RBA_ERR_INVALID_ARGUMENT;
    }
  }
  if (options->implicit_q == 0) {
    static bool warned = false;
    if (!warned)
      std::fprintf(stderr, "[rootba_hip] rba_options.implicit_q = 0 is ignored: the dense-block products were removed in "
                           "round 3, H*x is always evaluated from the factors\n");
    warned = true;
  }
  return RBA_OK;
}

// one solver on one device for the landmarks [lm0, lm1) of the caller's problem
rba_solver* make_solver(int dtype, int device, int32_t n_cams, int32_t lm0, int32_t lm1, const int64_t* lm_obs_offsets,
                        const int32_t* obs_cam_idx, const void* obs_xy, const rba_options& options) {
  const int32_t n_lms = lm1 - lm0;
  const int64_t o0 = lm_obs_offsets[lm0];
  std::vector<int64_t> off;
  const int64_t* offp = lm_obs_offsets;
  if (lm0 != 0) {  // (a shard: offsets relative to its first observation)
    off.resize(size_t(n_lms) + 1);
    for (int32_t l = 0; l <= n_lms; ++l) off[l] = lm_obs_offsets[lm0 + l] - o0;
    offp = off.data();
  }
  const int32_t* cam = obs_cam_idx + o0;
  if (dtype == RBA_F32) return new Solver<float>(device, n_cams, n_lms, offp, cam, static_cast<const float*>(obs_xy) + 2 * o0, options);
  if (dtype == RBA_F64) return new Solver<double>(device, n_cams, n_lms, offp, cam, static_cast<const double*>(obs_xy) + 2 * o0, options);
  if (dtype == RBA_MIXED) {
    const double* xy64 = static_cast<const double*>(obs_xy) + 2 * o0;
    std::vector<float> xy32(size_t(2) * size_t(lm_obs_offsets[lm1] - o0));
    for (size_t i = 0; i < xy32.size(); ++i) xy32[i] = float(xy64[i]);
    return new Solver<float>(device, n_cams, n_lms, offp, cam, xy32.data(), options, xy64);
  }
  throw HipError{"dtype must be RBA_F32, RBA_F64 or RBA_MIXED", RBA_ERR_INVALID_ARGUMENT};
}

// ---------------------------------------------------------------------------
// ONE handle over several devices of ONE process (rba_create_sharded; SURVEY.md 8b: "..., int n_gpus, handle*").
// The reference is one process (linearizor.cpp:133-150 builds ONE Linearizor for the whole problem), so the drop-in
// path can only reach more than one GPU if the library shards by itself: contiguous landmark ranges balanced by THIS
// layout's bytes per landmark, one Solver per device, one host thread per Solver (the Solvers' collectives have to be
// entered concurrently), every call of the C ABI fanned out to all of them. The ranks' replicated results (cost sums,
// b, increments, LM decisions) are bit-identical by construction (fixed-order reductions on all-reduced data), so the
// handle reports rank 0's.
// Transport: RCCL (one communicator over the devices, ncclCommInitRank from the rank's own thread) when the devices are
// distinct; devices that repeat (a single-GPU test box) exchange through host memory instead - an in-process all-reduce
// behind the callback transport (every rank copies in, rank order is the summation order, everybody copies out).
// ---------------------------------------------------------------------------
class ShardedSolver final : public rba_solver {
 public:
  ShardedSolver(int dtype, int n, const int* devices, int32_t n_cams, int32_t n_lms, const int64_t* lm_off,
                const int32_t* obs_cam, const void* obs_xy, const rba_options& opt)
      : n_(n), n_cams_(n_cams), es_state_(dtype == RBA_F32 ? 4 : 8), es_vec_(dtype == RBA_F64 ? 8 : 4) {
    // ---- landmark ranges: bytes a landmark of k observations moves per LM iteration in this layout ~ 120 k + 100
    //      (rows, records and reflectors per observation + the landmark's own records), never an empty range
    {
      std::vector<double> w(size_t(n_lms) + 1, 0.0);
      for (int32_t l = 0; l < n_lms; ++l) w[l + 1] = w[l] + 120.0 * double(lm_off[l + 1] - lm_off[l]) + 100.0;
      cuts_.assign(size_t(n) + 1, 0);
      for (int r = 1; r < n; ++r) {
        const int32_t c = int32_t(std::lower_bound(w.begin(), w.end(), w[n_lms] * r / n) - w.begin());
        cuts_[r] = std::min(std::max(c, cuts_[r - 1] + 1), n_lms - (n - r));
      }
      cuts_[n] = n_lms;
    }
    bool distinct = true;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < i; ++j) distinct &= devices[i] != devices[j];
    rccl_ = distinct && n > 1;
    Rccl::UniqueId uid{};
    if (rccl_) {
      if (!g_rccl.load()) throw HipError{"librccl.so not found: rba_create_sharded over distinct devices needs RCCL", RBA_ERR_HIP};
      if (g_rccl.GetUniqueId(&uid) != 0) throw HipError{"ncclGetUniqueId failed", RBA_ERR_HIP};
    }
    ranks_.resize(n);
    hctx_.resize(n);
    for (int r = 0; r < n; ++r) hctx_[r] = HostCtx{this, r};
    hbufs_.assign(n, nullptr);
    workers_.reserve(n);
    for (int r = 0; r < n; ++r) workers_.emplace_back([this, r] { worker(r); });
    try {
      // two phases: every Solver first - a rank whose construction fails (out of memory, a HIP error)
      // must not leave its peers waiting inside ncclCommInitRank or a collective of the structure union
      run_all([&](int r) {
        ranks_[r].reset(make_solver(dtype, devices[r], n_cams, cuts_[r], cuts_[r + 1], lm_off, obs_cam, obs_xy, opt));
      });
      if (n_ > 1)
        run_all([&](int r) {
          if (rccl_)
            ranks_[r]->comm_init(r, n_, &uid);
          else
            ranks_[r]->comm_init_callback(r, n_, &ShardedSolver::host_allreduce, &hctx_[r]);
        });
    } catch (...) {
      try {
        run_all([&](int r) { ranks_[r].reset(); }, /*even_if_broken=*/true);  // (on the ranks' own threads, like the destructor)
      } catch (...) {
      }
      stop_workers();
      throw;
    }
  }
  ~ShardedSolver() override {
    try {
      run_all([&](int r) { ranks_[r].reset(); }, /*even_if_broken=*/true);
    } catch (...) {
    }
    stop_workers();
  }

  int n_ranks() const { return n_; }
  const std::vector<int32_t>& cuts() const { return cuts_; }

  void comm_init(int, int, const void*) override {
    throw HipError{"a sharded handle owns its communicator: rba_comm_init does not apply", RBA_ERR_UNSUPPORTED};
  }
  void comm_init_callback(int, int, rba_allreduce_fn, void*) override {
    throw HipError{"a sharded handle owns its communicator: rba_comm_init_callback does not apply", RBA_ERR_UNSUPPORTED};
  }
  void comm_info(int* rank, int* nranks, int* transport) override { ranks_[0]->comm_info(rank, nranks, transport); }
  void comm_stats(int64_t* calls, int64_t* bytes, double* seconds) override { ranks_[0]->comm_stats(calls, bytes, seconds); }
  void set_state(const void* cams, const void* lms) override {
    run_all([&](int r) { ranks_[r]->set_state(cams, static_cast<const char*>(lms) + size_t(3) * cuts_[r] * es_state_); });
  }
  void get_state(void* cams, void* lms) override {
    std::vector<char> scratch(size_t(10) * n_cams_ * es_state_ * (n_ > 1 ? n_ - 1 : 0));
    run_all([&](int r) {
      void* c = r == 0 ? cams : static_cast<void*>(scratch.data() + size_t(r - 1) * 10 * n_cams_ * es_state_);
      ranks_[r]->get_state(c, static_cast<char*>(lms) + size_t(3) * cuts_[r] * es_state_);
    });
  }
  void backup() override { run_all([&](int r) { ranks_[r]->backup(); }); }
  void restore() override { run_all([&](int r) { ranks_[r]->restore(); }); }
  void compute_error(rba_residual_info* out) override {
    std::vector<rba_residual_info> o(n_);
    run_all([&](int r) { ranks_[r]->compute_error(&o[r]); });
    *out = o[0];
  }
  int linearize(void* jp_diag2_out) override {
    std::vector<char> scratch(jp_diag2_out ? size_t(9) * n_cams_ * es_vec_ * (n_ - 1) : 0);
    return run_all_status([&](int r) {
      void* o = !jp_diag2_out ? nullptr : r == 0 ? jp_diag2_out : static_cast<void*>(scratch.data() + size_t(r - 1) * 9 * n_cams_ * es_vec_);
      return ranks_[r]->linearize(o);
    });
  }
  int solve(double lambda, void* inc_out, rba_cg_summary* cg) override {
    std::vector<char> scratch(inc_out ? size_t(9) * n_cams_ * es_vec_ * (n_ - 1) : 0);
    std::vector<rba_cg_summary> c(n_);
    const int st = run_all_status([&](int r) {
      void* o = !inc_out ? nullptr : r == 0 ? inc_out : static_cast<void*>(scratch.data() + size_t(r - 1) * 9 * n_cams_ * es_vec_);
      return ranks_[r]->solve(lambda, o, &c[r]);
    });
    if (cg) *cg = c[0];
    return st;
  }
  int stage2(double lambda, void* b_out, void* blocks_out) override {
    std::vector<char> sb(b_out ? size_t(9) * n_cams_ * es_vec_ * (n_ - 1) : 0), sk(blocks_out ? size_t(81) * n_cams_ * es_vec_ * (n_ - 1) : 0);
    return run_all_status([&](int r) {
      void* b = !b_out ? nullptr : r == 0 ? b_out : static_cast<void*>(sb.data() + size_t(r - 1) * 9 * n_cams_ * es_vec_);
      void* k = !blocks_out ? nullptr : r == 0 ? blocks_out : static_cast<void*>(sk.data() + size_t(r - 1) * 81 * n_cams_ * es_vec_);
      return ranks_[r]->stage2(lambda, b, k);
    });
  }
  void right_multiply(const void* x, void* y) override {
    std::vector<char> scratch(size_t(9) * n_cams_ * es_vec_ * (n_ - 1));
    run_all([&](int r) { ranks_[r]->right_multiply(x, r == 0 ? y : static_cast<void*>(scratch.data() + size_t(r - 1) * 9 * n_cams_ * es_vec_)); });
  }
  void right_multiply_explicit(const void* x, void* y) override {
    std::vector<char> scratch(size_t(9) * n_cams_ * es_vec_ * (n_ - 1));
    run_all([&](int r) {
      ranks_[r]->right_multiply_explicit(x, r == 0 ? y : static_cast<void*>(scratch.data() + size_t(r - 1) * 9 * n_cams_ * es_vec_));
    });
  }
  int apply(const void* inc, double* l_diff, bool update_cams) override {
    std::vector<double> ld(n_, 0.0);
    const int st = run_all_status([&](int r) { return ranks_[r]->apply(inc, &ld[r], update_cams); });
    if (l_diff) *l_diff = ld[0];
    return st;
  }
  int optimize_lm(rba_lm_iteration* log, int max_rows, int* n_rows, int* term) override {
    std::vector<std::vector<rba_lm_iteration>> logs(n_, std::vector<rba_lm_iteration>(size_t(std::max(max_rows, 1))));
    std::vector<int> nr(n_, 0), tm(n_, 0);
    const int st = run_all_status([&](int r) { return ranks_[r]->optimize_lm(logs[r].data(), max_rows, &nr[r], &tm[r]); });
    for (int i = 0; i < std::min(nr[0], max_rows); ++i) log[i] = logs[0][i];
    *n_rows = nr[0];
    *term = tm[0];
    return st;
  }
  void lm_begin() override { run_all([&](int r) { ranks_[r]->lm_begin(); }); }
  int lm_step(rba_lm_iteration* out) override {
    std::vector<rba_lm_iteration> rows(n_);
    std::vector<int> more(n_, 0);
    run_all([&](int r) { more[r] = ranks_[r]->lm_step(&rows[r]); });
    *out = rows[0];
    return more[0];
  }
  int lm_termination() const override { return ranks_[0]->lm_termination(); }
  void device_sync() override { run_all([&](int r) { ranks_[r]->device_sync(); }); }
  int64_t debug_read_A(int vec) override {
    std::vector<int64_t> b(n_, 0);
    run_all([&](int r) { b[r] = ranks_[r]->debug_read_A(vec); });
    return std::accumulate(b.begin(), b.end(), int64_t(0));
  }
  void get_timings(rba_iter_timings* out) override { ranks_[0]->get_timings(out); }
  void get_substage_timings(rba_substage_timings* out) override { ranks_[0]->get_substage_timings(out); }
  void get_jl_col_scale(void* out) override {
    run_all([&](int r) { ranks_[r]->get_jl_col_scale(static_cast<char*>(out) + size_t(3) * cuts_[r] * es_vec_); });
  }
  void get_pose_scaling(void* out) override {
    std::vector<char> scratch(size_t(9) * n_cams_ * es_vec_ * (n_ - 1));
    run_all([&](int r) { ranks_[r]->get_pose_scaling(r == 0 ? out : static_cast<void*>(scratch.data() + size_t(r - 1) * 9 * n_cams_ * es_vec_)); });
  }
  void get_landmark_R(int damped, void* R6, void* q3) override {
    run_all([&](int r) {
      ranks_[r]->get_landmark_R(damped, static_cast<char*>(R6) + size_t(6) * cuts_[r] * es_vec_,
                                static_cast<char*>(q3) + size_t(3) * cuts_[r] * es_vec_);
    });
  }
  void get_landmark_q2tr_norm(void* out) override {
    run_all([&](int r) { ranks_[r]->get_landmark_q2tr_norm(static_cast<char*>(out) + size_t(cuts_[r]) * es_vec_); });
  }
  void get_problem_stats(int64_t* storage, int64_t* hx_bytes, int64_t* hx_flops) override {
    std::vector<int64_t> a(n_), b(n_), c(n_);
    run_all([&](int r) { ranks_[r]->get_problem_stats(&a[r], &b[r], &c[r]); });
    *storage = std::accumulate(a.begin(), a.end(), int64_t(0));
    *hx_bytes = std::accumulate(b.begin(), b.end(), int64_t(0));
    *hx_flops = std::accumulate(c.begin(), c.end(), int64_t(0));
  }
  void get_byte_model(rba_byte_model* out) override { ranks_[0]->get_byte_model(out); }  // (rank 0's shard)
  void get_pcg_counters(rba_pcg_counters* out) override { ranks_[0]->get_pcg_counters(out); }
  void get_reduced_matrix_info(rba_reduced_matrix_info* out) override { ranks_[0]->get_reduced_matrix_info(out); }

 private:
  // ---- one host thread per rank: the Solvers' collectives must be entered by all ranks at once --------------------
  void worker(int r) {
    uint64_t seen = 0;
    for (;;) {
      std::function<void(int)> f;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_job_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        f = job_;
      }
      try {
        f(r);
      } catch (const HipError& e) {
        fail(r, e);
      } catch (const std::exception& e) {
        fail(r, HipError{e.what(), RBA_ERR_HIP});
      }
      {
        std::lock_guard<std::mutex> lk(m_);
        if (--busy_ == 0) cv_done_.notify_all();
      }
    }
  }
  // A rank threw while its peers may be inside a collective that now never completes: record the first
  // error, wake the waiters of the host transport (they return an error), abort the peers' RCCL communicators (their
  // pending collectives return), and mark the handle broken - every later call fails at once instead of hanging.
  void fail(int r, const HipError& e) {
    {
      std::lock_guard<std::mutex> lk(m_);
      if (!failed_) err_ = e;
      failed_ = true;
      broken_ = true;
    }
    {
      std::lock_guard<std::mutex> lk(hm_);
      habort_ = true;
    }
    hcv_.notify_all();
    if (rccl_)
      for (int q = 0; q < n_; ++q)
        if (q != r && ranks_[q]) ranks_[q]->comm_abort();
  }
  void run_all(const std::function<void(int)>& f, bool even_if_broken = false) {
    {
      std::lock_guard<std::mutex> lk(m_);
      if (broken_ && !even_if_broken)
        throw HipError{"this sharded handle is broken: a rank failed earlier (" + err_.msg + ")", err_.code ? err_.code : RBA_ERR_HIP};
      job_ = f;
      busy_ = n_;
      failed_ = false;
      ++gen_;
    }
    cv_job_.notify_all();
    std::unique_lock<std::mutex> lk(m_);
    cv_done_.wait(lk, [&] { return busy_ == 0; });
    if (failed_) throw err_;
  }
  int run_all_status(const std::function<int(int)>& f) {
    std::vector<int> st(n_, RBA_OK);
    run_all([&](int r) { st[r] = f(r); });
    return st[0];  // (the ranks agree: failure flags are max-reduced)
  }
  void stop_workers() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_job_.notify_all();
    for (auto& t : workers_)
      if (t.joinable()) t.join();
  }
  // ---- all-reduce through host memory for ranks that share a device (rba_allreduce_fn) -------------------------------
  struct HostCtx {
    ShardedSolver* self;
    int rank;
  };
  static int host_allreduce(void* ctx, void* buf, int64_t count, int dtype, int op) {
    const HostCtx* c = static_cast<const HostCtx*>(ctx);
    return c->self->host_allreduce_impl(c->rank, buf, count, dtype, op);
  }
  template <class T>
  static void reduce_into(T* dst, const T* src, int64_t n, int op) {
    if (op == 0)
      for (int64_t i = 0; i < n; ++i) dst[i] += src[i];
    else
      for (int64_t i = 0; i < n; ++i) dst[i] = std::max(dst[i], src[i]);
  }
  int host_allreduce_impl(int rank, void* buf, int64_t count, int dtype, int op) {
    const size_t es = dtype == 1 ? 8 : 4;
    std::unique_lock<std::mutex> lk(hm_);
    if (habort_) return 1;
    const uint64_t my_gen = hgen_;
    hbufs_[rank] = buf;
    if (++harrived_ == n_) {
      // the last one in reduces, in rank order (the summation order of every call), and hands the result to everybody
      hacc_.resize(size_t(count) * es);
      std::memcpy(hacc_.data(), hbufs_[0], size_t(count) * es);
      for (int r = 1; r < n_; ++r) {
        if (dtype == 0) reduce_into(reinterpret_cast<float*>(hacc_.data()), static_cast<const float*>(hbufs_[r]), count, op);
        else if (dtype == 1) reduce_into(reinterpret_cast<double*>(hacc_.data()), static_cast<const double*>(hbufs_[r]), count, op);
        else reduce_into(reinterpret_cast<int*>(hacc_.data()), static_cast<const int*>(hbufs_[r]), count, op);
      }
      for (int r = 0; r < n_; ++r) std::memcpy(hbufs_[r], hacc_.data(), size_t(count) * es);
      harrived_ = 0;
      ++hgen_;
      hcv_.notify_all();
    } else {
      hcv_.wait(lk, [&] { return hgen_ != my_gen || habort_; });
      if (hgen_ == my_gen) return 1;  // (aborted: a sibling rank failed and will never arrive)
    }
    return 0;
  }

  int n_, n_cams_;
  size_t es_state_, es_vec_;
  bool rccl_ = false;
  std::vector<int32_t> cuts_;
  std::vector<std::unique_ptr<rba_solver>> ranks_;
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_job_, cv_done_;
  std::function<void(int)> job_;
  uint64_t gen_ = 0;
  int busy_ = 0;
  bool stop_ = false, failed_ = false, broken_ = false, habort_ = false;
  HipError err_{"", 0};
  std::mutex hm_;
  std::condition_variable hcv_;
  std::vector<HostCtx> hctx_;
  std::vector<void*> hbufs_;
  std::vector<char> hacc_;
  int harrived_ = 0;
  uint64_t hgen_ = 0;
};

}  // namespace

extern "C" {

void rba_default_options(rba_options* o) {
  o->use_householder = 1;
  o->use_valid_projections_only = 0;
  o->robust_norm = 0;
  o->huber_parameter = 1.0;
  o->jacobi_scaling_eps = 0.0;
  o->preconditioner_type = 1;
  o->reduction_alg = 1;
  o->power_order = 10;
  o->min_cg_it = 0;
  o->max_cg_it = 500;
  o->eta = 0.1;
  o->num_threads = 0;
  o->max_num_iterations = 20;
  o->min_relative_decrease = 0.0;
  o->initial_trust_region_radius = 1e4;
  o->min_trust_region_radius = 1e-32;
  o->max_trust_region_radius = 1e16;
  o->function_tolerance = 1e-6;
  o->initial_vee = 2.0;
  o->vee_factor = 2.0;
  o->optimized_cost = 0;
  o->staged_execution = 1;
  o->implicit_q = 1;
  o->solver_type = 0;
  o->explicit_after = -1;
}

const char* rba_last_error(void) { return g_last_error.c_str(); }

int rba_device_count(int* out) {
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    g_last_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
    *out = 0;
    return RBA_ERR_HIP;
  }
  *out = n;
  return RBA_OK;
}


int rba_create(int dtype, int device, int32_t n_cams, int32_t n_lms,
               const int64_t* lm_obs_offsets, const int32_t* obs_cam_idx, const void* obs_xy,
               const rba_options* options, rba_handle* out) {
  return guarded([&]() -> int {
    if (const int st = validate_create("rba_create", n_cams, n_lms, lm_obs_offsets, obs_cam_idx, obs_xy, options, out)) return st;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      g_last_error = "no HIP device available: the solver has no CPU fallback";
      return RBA_ERR_HIP;
    }
    if (device < 0 || device >= ndev) {
      g_last_error = "device index out of range";
      return RBA_ERR_INVALID_ARGUMENT;
    }
    *out = make_solver(dtype, device, n_cams, 0, n_lms, lm_obs_offsets, obs_cam_idx, obs_xy, *options);
    return RBA_OK;
  });
}

int rba_create_sharded(int dtype, int n_gpus, const int* device_ids, int32_t n_cams, int32_t n_lms,
                       const int64_t* lm_obs_offsets, const int32_t* obs_cam_idx, const void* obs_xy,
                       const rba_options* options, rba_handle* out) {
  return guarded([&]() -> int {
    if (const int st = validate_create("rba_create_sharded", n_cams, n_lms, lm_obs_offsets, obs_cam_idx, obs_xy, options, out))
      return st;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      g_last_error = "no HIP device available: the solver has no CPU fallback";
      return RBA_ERR_HIP;
    }
    if (n_gpus < 1 || n_gpus > 64 || n_gpus > n_lms) {
      g_last_error = "rba_create_sharded: 1 <= n_gpus <= min(64, n_lms) required";
      return RBA_ERR_INVALID_ARGUMENT;
    }
    std::vector<int> dev(n_gpus);
    for (int i = 0; i < n_gpus; ++i) {
      dev[i] = device_ids ? device_ids[i] : i;  // (null: devices 0 .. n_gpus - 1)
      if (dev[i] < 0 || dev[i] >= ndev) {
        g_last_error = "rba_create_sharded: device index out of range";
        return RBA_ERR_INVALID_ARGUMENT;
      }
    }
    if (dtype != RBA_F32 && dtype != RBA_F64 && dtype != RBA_MIXED) {
      g_last_error = "dtype must be RBA_F32, RBA_F64 or RBA_MIXED";
      return RBA_ERR_INVALID_ARGUMENT;
    }
    *out = new ShardedSolver(dtype, n_gpus, dev.data(), n_cams, n_lms, lm_obs_offsets, obs_cam_idx, obs_xy, *options);
    return RBA_OK;
  });
}

int rba_get_shard_ranges(rba_handle h, int* n_ranks_out, int32_t* cuts_out, int max_cuts) {
  return guarded([&]() -> int {
    auto* s = dynamic_cast<ShardedSolver*>(h);
    if (!s) {
      if (n_ranks_out) *n_ranks_out = 1;
      return RBA_OK;
    }
    if (n_ranks_out) *n_ranks_out = s->n_ranks();
    if (cuts_out)
      for (int i = 0; i < std::min<int>(max_cuts, int(s->cuts().size())); ++i) cuts_out[i] = s->cuts()[i];
    return RBA_OK;
  });
}

int rba_destroy(rba_handle h) {
  return guarded([&]() -> int {
    delete h;
    return RBA_OK;
  });
}

int rba_comm_unique_id(void* out128) {
  return guarded([&]() -> int {
    if (!g_rccl.load()) {
      g_last_error = "cannot load librccl.so";
      return RBA_ERR_COMM;
    }
    Rccl::UniqueId id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) {
      g_last_error = "ncclGetUniqueId failed";
      return RBA_ERR_COMM;
    }
    std::memcpy(out128, &id, sizeof(id));
    return RBA_OK;
  });
}

int rba_comm_init(rba_handle h, int rank, int nranks, const void* uid) {
  return guarded([&]() -> int {
    h->comm_init(rank, nranks, uid);
    return RBA_OK;
  });
}

int rba_comm_init_callback(rba_handle h, int rank, int nranks, rba_allreduce_fn fn, void* ctx) {
  return guarded([&]() -> int {
    if (nranks > 1 && !fn) {
      g_last_error = "rba_comm_init_callback: null callback";
      return RBA_ERR_INVALID_ARGUMENT;
    }
    h->comm_init_callback(rank, nranks, fn, ctx);
    return RBA_OK;
  });
}

int rba_comm_info(rba_handle h, int* rank_out, int* nranks_out, int* transport_out) {
  return guarded([&]() -> int {
    h->comm_info(rank_out, nranks_out, transport_out);
    return RBA_OK;
  });
}
int rba_get_comm_stats(rba_handle h, int64_t* calls_out, int64_t* bytes_out, double* seconds_out) {
  return guarded([&]() -> int {
    h->comm_stats(calls_out, bytes_out, seconds_out);
    return RBA_OK;
  });
}

int rba_set_state(rba_handle h, const void* cams, const void* lms) {
  return guarded([&]() -> int {
    h->set_state(cams, lms);
    return RBA_OK;
  });
}
int rba_get_state(rba_handle h, void* cams, void* lms) {
  return guarded([&]() -> int {
    h->get_state(cams, lms);
    return RBA_OK;
  });
}
int rba_backup(rba_handle h) {
  return guarded([&]() -> int {
    h->backup();
    return RBA_OK;
  });
}
int rba_restore(rba_handle h) {
  return guarded([&]() -> int {
    h->restore();
    return RBA_OK;
  });
}
int rba_compute_error(rba_handle h, rba_residual_info* out) {
  return guarded([&]() -> int {
    h->compute_error(out);
    return RBA_OK;
  });
}
int rba_linearize(rba_handle h, void* jp_diag2_out) {
  return guarded([&]() -> int { return h->linearize(jp_diag2_out); });
}
int rba_solve(rba_handle h, double lambda, void* inc_out, rba_cg_summary* cg) {
  return guarded([&]() -> int {
    if (!inc_out) throw HipError{"rba_solve: inc_out is NULL", RBA_ERR_INVALID_ARGUMENT};
    return h->solve(lambda, inc_out, cg);
  });
}
int rba_stage2(rba_handle h, double lambda, void* b_out, void* blocks_out) {
  return guarded([&]() -> int { return h->stage2(lambda, b_out, blocks_out); });
}
int rba_right_multiply(rba_handle h, const void* x, void* y) {
  return guarded([&]() -> int {
    h->right_multiply(x, y);
    return RBA_OK;
  });
}

int rba_right_multiply_explicit(rba_handle h, const void* x, void* y) {
  return guarded([&]() -> int {
    h->right_multiply_explicit(x, y);
    return RBA_OK;
  });
}
int rba_apply(rba_handle h, const void* inc, double* l_diff_out) {
  return guarded([&]() -> int {
    if (!inc) throw HipError{"rba_apply: inc is NULL", RBA_ERR_INVALID_ARGUMENT};
    return h->apply(inc, l_diff_out, true);
  });
}
int rba_back_substitute(rba_handle h, const void* inc, double* l_diff_out) {
  return guarded([&]() -> int {
    if (!inc) throw HipError{"rba_back_substitute: inc is NULL", RBA_ERR_INVALID_ARGUMENT};
    return h->apply(inc, l_diff_out, false);
  });
}
int rba_optimize_lm(rba_handle h, rba_lm_iteration* log, int max_rows, int* n_rows_out,
                    int* termination_out) {
  return guarded([&]() -> int { return h->optimize_lm(log, max_rows, n_rows_out, termination_out); });
}
int rba_lm_begin(rba_handle h) {
  return guarded([&]() -> int {
    h->lm_begin();
    return RBA_OK;
  });
}
int rba_lm_step(rba_handle h, rba_lm_iteration* row, int* more_out) {
  return guarded([&]() -> int {
    const int more = h->lm_step(row);
    if (more_out) *more_out = more;
    return RBA_OK;
  });
}
int rba_lm_termination(rba_handle h, int* termination_out) {
  return guarded([&]() -> int {
    *termination_out = h->lm_termination();
    return RBA_OK;
  });
}
int rba_synchronize(rba_handle h) {
  return guarded([&]() -> int {
    h->device_sync();
    return RBA_OK;
  });
}
int rba_debug_read_blocks(rba_handle h, int vec_width, int64_t* bytes_out) {
  return guarded([&]() -> int {
    *bytes_out = h->debug_read_A(vec_width);
    return RBA_OK;
  });
}
int rba_get_timings(rba_handle h, rba_iter_timings* out) {
  return guarded([&]() -> int {
    h->get_timings(out);
    return RBA_OK;
  });
}
int rba_get_substage_timings(rba_handle h, rba_substage_timings* out) {
  return guarded([&]() -> int {
    if (!out) return RBA_ERR_INVALID_ARGUMENT;
    h->get_substage_timings(out);
    return RBA_OK;
  });
}
int rba_get_jl_col_scale(rba_handle h, void* out) {
  return guarded([&]() -> int {
    h->get_jl_col_scale(out);
    return RBA_OK;
  });
}
int rba_get_pose_scaling(rba_handle h, void* out) {
  return guarded([&]() -> int {
    h->get_pose_scaling(out);
    return RBA_OK;
  });
}
int rba_get_landmark_R(rba_handle h, int damped, void* R6, void* q3) {
  return guarded([&]() -> int {
    h->get_landmark_R(damped, R6, q3);
    return RBA_OK;
  });
}
int rba_get_landmark_q2tr_norm(rba_handle h, void* out) {
  return guarded([&]() -> int {
    if (!out) return RBA_ERR_INVALID_ARGUMENT;
    h->get_landmark_q2tr_norm(out);
    return RBA_OK;
  });
}
int rba_get_byte_model(rba_handle h, rba_byte_model* out) {
  return guarded([&]() -> int {
    h->get_byte_model(out);
    return RBA_OK;
  });
}
int rba_get_pcg_counters(rba_handle h, rba_pcg_counters* out) {
  return guarded([&]() -> int {
    if (!out) return RBA_ERR_INVALID_ARGUMENT;
    h->get_pcg_counters(out);
    return RBA_OK;
  });
}
int rba_get_reduced_matrix_info(rba_handle h, rba_reduced_matrix_info* out) {
  return guarded([&]() -> int {
    if (!out) return RBA_ERR_INVALID_ARGUMENT;
    h->get_reduced_matrix_info(out);
    return RBA_OK;
  });
}
int rba_get_problem_stats(rba_handle h, int64_t* storage, int64_t* hx_bytes, int64_t* hx_flops) {
  return guarded([&]() -> int {
    h->get_problem_stats(storage, hx_bytes, hx_flops);
    return RBA_OK;
  });
}

}  // extern "C"
