// kernels.hpp — hand-written HIP kernels (gfx950) of the square-root BA solver.
//
// Work runs at the parallelism it has: thread per observation (kernels_s1.hpp), lane per block row
// on WAVE TILES (a landmark = an aligned group of 4..64 lanes: k_hx_implicit*, k_bs_tile, k_s1_qr_tile), workgroup
// per camera for the camera-indexed sums (kernels_cam.hpp), workgroup per landmark for tracks longer than 112
// (kernels_big.hpp). Products scatter into workgroup-private double copies of y in LDS (k_hx_implicit_lds).
// No dense landmark block exists: the reference's (2k+3) x (9k+pad+4) LandmarkBlock storage
// (landmark_block_dynamic.hpp:49-69) is replaced by per-row records (Jacobian rows, Householder vectors) and
// per-landmark scalars; the dense-block kernel generation of rounds 1-2 (one wavefront per landmark along the 9k
// columns) was removed in round 3.
// Landmarks are sorted by track-length class and by first camera inside a class at setup, so a launch covers one
// contiguous class. See DESIGN.md for bytes per kernel.
#pragma once

#include <type_traits>

#include "device_utils.hpp"

namespace rba {

// Per-observation record of damped top rows [damped Q1^T Jp D 3x9 | 5 unused] = 32 scalars: one 128-byte cache line
// per record in float, so the pair gather of the reduced-matrix assembly fetches exactly the lines it needs (round
// 2: 36 scalars incl. a b part nobody reads any more - 144-byte records straddled lines, 6.8 GB fetched per
// venice assembly for 3.7 GB of rows).
constexpr int kTd = 32;

template <class S>
struct Params {
  int n_cams;
  int n_lms;
  // topology (sorted landmark order)
  const int* __restrict__ lm_k;         // [n_lms]
  const int64_t* __restrict__ lm_obs;   // [n_lms+1] first observation
  const int* __restrict__ obs_cam;      // [n_obs]
  const int* __restrict__ obs_lm;       // [n_obs] sorted landmark index
  const S* __restrict__ obs_xy;         // [2 n_obs]
  const int64_t* __restrict__ cam_obs_off;  // [n_cams+1] CSC: observations of a camera
  const int* __restrict__ cam_obs;          // [n_obs] sorted-observation indices
  // state
  S* cams;  // [10 n_cams]
  S* lms;   // [3 n_lms]
  // per-observation / per-row records (landmark-major)
  S* JpS;       // [2 n_obs][8]  weighted pose Jacobian rows, entries 0..7 of block row w = 2 o + r, UNSCALED: the Jacobi
                //               column scaling D (pose_scaling) is applied where a camera index is at hand (operand and
                //               result of H x, increment of the back-substitution, epilogue of the camera-major pass)
  S* JpT;       // [2 n_obs]     ... and entry 8 of every row (jp_row / JpRows below: why the rows are split)
  S* Vh;        // [2 n_obs][4]  per block row: Householder vectors v0, v1, v2 and (Q^T r)[row]
  S* JlS;       // [n_obs][2][3] sqrt(w) Jl D_l before the QR (back-substitution)
  S* rS;        // [n_obs][2]    sqrt(w) r
  S* WA;        // [n_obs][8]    stage-2 record of the camera-major pass (kernels_cam.hpp): [g 2 | A 2x2 | 0 0] with the
                //               observation's part of b = (Jp D)^T g and of the diagonal block = (A Jp D)^T (A Jp D)
  S* W8;             // [n_obs - w8_begin][8] the eight stage-2 coefficients W' (3x2, row-major) | g (2) themselves, kept
  int64_t w8_begin;  //   only for the observations of the two-kernel back-substitution (k > 32): topd x = W' (Jp D x)
  S* topd;      // [n_obs][kTd]  damped Q1^T Jp D ([3][9], padded to 32): materialised ON DEMAND from
                //               JpS and the factors (k_s12_cols) for the assembly of the reduced matrix and E0 products
  S* bsO;       // [n_obs][5]    back-substitution scratch (k > 32): topd x (3), Jp x (2)
  // per-landmark records
  S* tauH;      // [3 n_lms]     reflector tau
  S* LQ;        // [n_lms][12]   tau[3], reflector cross products g10 g20 g21, d[3] (kernels_s1.hpp)
  S* R0;        // [6 n_lms]     upper 3x3 of R, undamped
  S* jl_scale;  // [3 n_lms]     Jl_col_scale
  S* givens;    // [n_lms][16]   the 6 damping rotations of stage 2: c[6], s[6], damping-row residual[3], pad
  S* Rd;        // [6 n_lms]     damped R
  S* q1trd;     // [3 n_lms]     damped Q1^T r
  S* damp_r;    // [3 n_lms]
  S* Zd;        // [9 n_lms]     3x3 map of the top rows through damp / drop-Q1 / undamp (implicit-Q operator)
  // wave tiles for k <= 32 (one lane per block row, landmark = aligned group of P2 lanes, see
  // k_hx_implicit / k_s1_qr_tile): static maps lane -> camera / block row, -1 = padding
  const int2* __restrict__ OT;       // [tiles][32] per OBSERVATION slot of a tile (lanes 2 q, 2 q + 1): {camera, global
                                     //              block row 2 o of its first row}; camera -1 = padding. (Rounds 1-5: two
                                     //              maps of an entry per LANE - the second lane's entries were the first's
                                     //              + {0, 1}: 16 bytes per observation in the product, the back-substitution
                                     //              and stage 1 where 8 say the same; tile_map below.)
  S* lm_inc;     // mixed precision (RBA_MIXED): the back-substitution stores the scaled landmark increments here
                 // [3 n_lms] instead of adding them to `lms`; they are applied to the double master state
  // camera-sized vectors
  S* jp_diag2;      // [9 n_cams]
  S* pose_scaling;  // [9 n_cams]
  S* B_mid;         // [81 n_cams] D G D (JACOBI blocks without the damping)
  S* b;             // [9 n_cams]
  S* blocks;        // [81 n_cams]
  S* sdiag;         // [81 n_cams] JACOBI / power-series with the assembled matrix: the diagonal blocks of the reduced
  int want_sdiag;   //             matrix (the preconditioner blocks are B_mid + lambda I there)
  int* fail_flag;   // numerical failure
  double* lm_ldiff;  // [n_lms] per-landmark model cost change
  // options
  int robust_norm;
  int valid_only;
  int jacobi;  // preconditioner_type == JACOBI / POWER_SCHUR_COMPLEMENT: blocks = Hpp + lambda I
  S huber;
  S eps;  // jacobi scaling epsilon
  // stage timers of rba_lm_step (solver.hip: time_begin / time_end): the FIRST kernel of a stage leaves the chip-wide
  // 100 MHz clock here when it starts (nullptr otherwise) - a HIP event per stage boundary is a marker packet of ~4 us
  // on the queue, 50 us per LM iteration (15 % of a trafalgar-257 iteration)
  unsigned long long* stamp;
};

// ---------------------------------------------------------------------------------------------------------------------
// Storage of the weighted pose-Jacobian rows, SPLIT since round 6: block row w = 2 o + r keeps its first eight entries
// at JpS + 8 w and its ninth at JpT[w]. The two rows of an observation are then ONE aligned 64-byte line (float) and
// the landmark-major kernels read the same 72 bytes per observation as before, as two perfectly coalesced streams
// (16-byte vectors of the main part, 4 bytes of the tail per lane). Rounds 1-5 kept 18 consecutive scalars per
// observation: a 72-byte record at a multiple of 72 bytes always straddles TWO 64-byte lines, and the camera-major pass,
// which gathers one record per observation of a camera, fetched 128 bytes of rows + 64 of its 32-byte stage-2 record
// for 104 it used - 1.86 x, measured (profiles/r5_pmc_stage_traffic.csv; VERDICT rounds 3-5). With the split the
// gather is one line of rows + one line that holds the stage-2 record AND the two tail entries (WA[6..7]): 128 bytes.
// ---------------------------------------------------------------------------------------------------------------------
// a lane's camera and block row from the tile map (-1, -1: padding)
template <class S>
__device__ __forceinline__ void tile_map(const Params<S>& p, size_t T, int lane, int& cam, int& row) {
  const int2 e = p.OT[T * 32 + (lane >> 1)];
  cam = e.x;
  row = e.x >= 0 ? e.y + (lane & 1) : -1;
}

template <class S>
__device__ __forceinline__ void jp_row(const S* __restrict__ J8, const S* __restrict__ JT, int64_t w, S (&jp)[9]) {
  using V4 = typename std::conditional<sizeof(S) == 4, float4, double4>::type;
  const V4* __restrict__ m = reinterpret_cast<const V4*>(J8) + 2 * w;
  const V4 a = m[0], b = m[1];
  jp[0] = a.x, jp[1] = a.y, jp[2] = a.z, jp[3] = a.w;
  jp[4] = b.x, jp[5] = b.y, jp[6] = b.z, jp[7] = b.w;
  jp[8] = JT[w];
}
// both rows of observation o: jp[0..8] row 0, jp[9..17] row 1
template <class S>
__device__ __forceinline__ void jp_obs(const S* __restrict__ J8, const S* __restrict__ JT, int64_t o, S (&jp)[18]) {
  using V4 = typename std::conditional<sizeof(S) == 4, float4, double4>::type;
  using V2 = typename std::conditional<sizeof(S) == 4, float2, double2>::type;
  const V4* __restrict__ m = reinterpret_cast<const V4*>(J8) + 4 * o;
  const V4 a = m[0], b = m[1], c = m[2], d = m[3];
  const V2 t = reinterpret_cast<const V2*>(JT)[o];
  jp[0] = a.x, jp[1] = a.y, jp[2] = a.z, jp[3] = a.w, jp[4] = b.x, jp[5] = b.y, jp[6] = b.z, jp[7] = b.w, jp[8] = t.x;
  jp[9] = c.x, jp[10] = c.y, jp[11] = c.z, jp[12] = c.w, jp[13] = d.x, jp[14] = d.y, jp[15] = d.z, jp[16] = d.w, jp[17] = t.y;
}
// `n` consecutive observations from o_base between an array [n][18] (row-major rows of nine, in LDS) and the split
// storage, by `nt` work-items with coalesced 16-byte accesses of the main part
template <class S>
__device__ __forceinline__ void jp_store_rows(S* __restrict__ J8, S* __restrict__ JT, int64_t o_base, int n,
                                              const S* __restrict__ rows18, int tid, int nt) {
  using V4 = typename std::conditional<sizeof(S) == 4, float4, double4>::type;
  V4* __restrict__ dst = reinterpret_cast<V4*>(J8) + 4 * o_base;
  for (int i = tid; i < 4 * n; i += nt) {  // piece i: observation i / 4, row (i / 2) & 1, entries 4 (i & 1) ...
    const S* __restrict__ src = rows18 + 18 * (i >> 2) + 9 * ((i >> 1) & 1) + 4 * (i & 1);
    dst[i] = V4{src[0], src[1], src[2], src[3]};
  }
  for (int i = tid; i < 2 * n; i += nt) JT[2 * o_base + i] = rows18[9 * i + 8];
}
template <class S>
__device__ __forceinline__ void jp_load_rows(const S* __restrict__ J8, const S* __restrict__ JT, int64_t o_base, int n,
                                             S* __restrict__ rows18, int tid, int nt) {
  using V4 = typename std::conditional<sizeof(S) == 4, float4, double4>::type;
  const V4* __restrict__ src = reinterpret_cast<const V4*>(J8) + 4 * o_base;
  for (int i = tid; i < 4 * n; i += nt) {
    const V4 v = src[i];
    S* __restrict__ dst = rows18 + 18 * (i >> 2) + 9 * ((i >> 1) & 1) + 4 * (i & 1);
    dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
  }
  for (int i = tid; i < 2 * n; i += nt) rows18[9 * i + 8] = JT[2 * o_base + i];
}

__device__ __forceinline__ void stage_stamp(unsigned long long* s) {
  if (s && blockIdx.x == 0 && threadIdx.x == 0) *s = wall_clock64();
}
// a stage boundary that no kernel takes with it (before a synchronisation)
__global__ void k_stamp(unsigned long long* s) { *s = wall_clock64(); }

// End of a landmark's back-substitution: model-cost term, non-finite check, and the update (ipp:279-283): lms +=
// Jl_col_scale * inc, or - mixed precision - the scaled increment is handed to the double master state
// (k_mixed_update_landmarks). All loads are issued before the
// first store and none sits behind a short-circuit: through pointers that may alias, a load behind a store is issued in
// order after it, and load - store pairs (or loads guarded by the result of earlier ones) in a row are as many memory
// round trips at the tail of every wave.
template <class S>
__device__ __forceinline__ void finish_landmark(const Params<S>& p, int s, const S inc[3], S acc) {
  const S* __restrict__ sc = p.jl_scale + 3 * size_t(s);
  const S* __restrict__ lm = p.lms + 3 * size_t(s);
  const S s0 = sc[0], s1 = sc[1], s2 = sc[2];
  const S l0 = lm[0], l1 = lm[1], l2 = lm[2];
  // (bit-wise on purpose: `&&` would put the later operands - and their loads - behind branches)
  const bool fin = (int(is_finite(inc[0])) & int(is_finite(inc[1])) & int(is_finite(inc[2])) & int(is_finite(acc)) &
                    int(is_finite(l0)) & int(is_finite(l1)) & int(is_finite(l2))) != 0;
  p.lm_ldiff[s] = -double(acc);
  if (!fin) atomicOr(p.fail_flag, 2);
  const S d0 = inc[0] * s0, d1 = inc[1] * s1, d2 = inc[2] * s2;
  if (p.lm_inc) {
    p.lm_inc[3 * size_t(s) + 0] = d0;
    p.lm_inc[3 * size_t(s) + 1] = d1;
    p.lm_inc[3 * size_t(s) + 2] = d2;
  } else {
    p.lms[3 * size_t(s) + 0] = l0 + d0;
    p.lms[3 * size_t(s) + 1] = l1 + d1;
    p.lms[3 * size_t(s) + 2] = l2 + d2;
  }
}

// What the end of an LM iteration hands to the host in ONE piece (rba_lm_step): the eight cost sums of the trial point,
// the model cost change of the back-substitution and the failure word - [0..7] cost sums, [8] l_diff, [9] failure bit 1
// (linearisation), [10] bit 2 (back-substitution), [11] bit 4 (block inversion) as 0 / 1 so that a SUM all-reduce of
// the twelve doubles is the collective of a sharded run (one instead of three).
constexpr int kEndRed = 12;

// Sum rows of a [n][W] double array into out[W] with the work-items of ONE workgroup (deterministic: work-item-strided,
// wavefronts in order). `sm`: [16][W] doubles of LDS.
template <int W>
__device__ __forceinline__ void reduce_rows_workgroup(const double* __restrict__ in, int64_t n, double* __restrict__ out,
                                                      double* __restrict__ out_host, double (*sm)[W]) {
  const int nt = int(blockDim.x), n_waves = nt >> 6;
  double acc[W];
#pragma unroll
  for (int i = 0; i < W; ++i) acc[i] = 0;
  for (int64_t r = threadIdx.x; r < n; r += nt) {
#pragma unroll
    for (int i = 0; i < W; ++i) acc[i] += in[r * W + i];
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < W; ++i) {
    const double t = wave_sum(acc[i]);
    if (lane == 0) sm[wave][i] = t;
  }
  __syncthreads();
  if (threadIdx.x < W) {
    double t = sm[0][threadIdx.x];
    for (int w = 1; w < n_waves; ++w) t += sm[w][threadIdx.x];
    out[threadIdx.x] = t;
    if (out_host) out_host[threadIdx.x] = t;
  }
}

// ===========================================================================
// compute_error: one thread per observation, per-block partial sums (double)
// (BalBundleAdjustmentHelper::compute_error, helper.cpp:68-109;
//  ResidualInfoAccu::add, residual_info.cpp:97-110)
// partials layout: [block][8] = {n_all, e_all, r_all, n_valid, e_valid, r_valid, n_nonfinite, 0}
// ===========================================================================
template <class S>
__global__ __launch_bounds__(256) void k_compute_error(Params<S> p, int64_t n_obs,
                                                       double* __restrict__ partials) {
  stage_stamp(p.stamp);
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  // (the two indices of the work-item's NEXT observation are requested before the current one is gathered: one memory
  //  round trip per observation instead of two dependent ones)
  const int64_t stride = gridDim.x * 256ll, o0 = blockIdx.x * 256ll + threadIdx.x;
  int cam = p.obs_cam[min(o0, n_obs - 1)], l = p.obs_lm[min(o0, n_obs - 1)];
  for (int64_t o = o0; o < n_obs; o += stride) {
    const int64_t on = min(o + stride, n_obs - 1);
    const int cam_next = p.obs_cam[on], l_next = p.obs_lm[on];
    S rx, ry;
    const bool valid = project_residual<S>(p.cams + 10 * cam, p.lms[3 * l], p.lms[3 * l + 1],
                                           p.lms[3 * l + 2], p.obs_xy[2 * o], p.obs_xy[2 * o + 1],
                                           rx, ry);
    const S r2 = rx * rx + ry * ry;
    S err, w;
    error_weight<S>(p.robust_norm, p.huber, r2, err, w);
    const double rn = double(sqrt(r2));
    acc[0] += 1.0;
    acc[1] += double(err);
    acc[2] += rn;
    if (valid) {
      acc[3] += 1.0;
      acc[4] += double(err);
      acc[5] += rn;
    }
    if (!(is_finite(rx) && is_finite(ry))) acc[6] += 1.0;
    cam = cam_next;
    l = l_next;
  }
  __shared__ double sm[4][7];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const double t = wave_sum(acc[i]);
    if (lane == 0) sm[wave][i] = t;
  }
  __syncthreads();
  if (threadIdx.x < 8)
    partials[blockIdx.x * 8 + threadIdx.x] =
        threadIdx.x < 7 ? sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x] : 0.0;
}

// Sum rows of a [n][W] double array into out[W] (single block, deterministic).
// Results that the host reads at its next synchronisation point are written straight into the pinned host page
// (`out_host`, `flag_host`: mapped, fine-grained) by the kernel that produces them - a device-to-host copy of a few
// bytes is a blit kernel of its own (4.8 us each on the profile, eight to ten per LM iteration). `flag`: the numerical
// failure word of the phase that ends here is published and its bits `flag_clear` are reset for the next phase.
// Null host pointers (more than one rank: the values are all-reduced on the device first) leave that to the caller.
// (any multiple of 64 work-items up to 1024. The cost evaluation's reduction runs with 256: with 1024 it waits for a
//  compute unit with sixteen free wave slots - on the side stream, beside the stage-1 kernels, 94 us instead of 12)
template <int W>
__global__ __launch_bounds__(1024) void k_reduce_rows(const double* __restrict__ in, int64_t n,
                                                      double* __restrict__ out, double* __restrict__ out_host,
                                                      int* __restrict__ flag, int* __restrict__ flag_host,
                                                      int flag_clear, unsigned long long* __restrict__ end_stamp,
                                                      double* __restrict__ bits = nullptr) {
  const int nt = int(blockDim.x);
  if ((flag_host || bits) && int(threadIdx.x) == nt - 1) {
    const int f = *flag;
    if (flag_host) *flag_host = f;
    if (bits) {  // rba_lm_step of a sharded run: the word's bits 1 / 2 / 4 as 0 / 1, summed over the ranks (kEndRed)
      bits[0] = (f & 1) ? 1.0 : 0.0;
      bits[1] = (f & 2) ? 1.0 : 0.0;
      bits[2] = (f & 4) ? 1.0 : 0.0;
    }
    if (f & flag_clear) *flag = f & ~flag_clear;
  }
  __shared__ double sm[16][W];
  reduce_rows_workgroup<W>(in, n, out, out_host, sm);
  if (end_stamp && threadIdx.x == 0) *end_stamp = wall_clock64();  // (the last kernel of its stage: see stage_stamp)
}

// the failure word of a phase that ends without a reduction of its own (stage 1)
__global__ void k_publish_flag(int* __restrict__ flag, int* __restrict__ flag_host, int flag_clear) {
  const int f = *flag;
  *flag_host = f;
  if (f & flag_clear) *flag = f & ~flag_clear;
}

// ===========================================================================
// Camera-major reductions. Every camera-indexed accumulator of the reference
// (Jp_diag2, b, the 9x9 block diagonal; reduction table in SURVEY.md §2) is
// produced by ONE workgroup per camera that walks the camera's observation list
// (CSC index built at setup) — no atomics, fixed summation order, double
// accumulators. The landmark-major kernels only write per-observation records.
// ===========================================================================
__device__ __forceinline__ double block_sum_256(double v, double* sm4) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double t = wave_sum(v);
  __syncthreads();
  if (lane == 0) sm4[wave] = t;
  __syncthreads();
  return sm4[0] + sm4[1] + sm4[2] + sm4[3];
}

// f32 accumulator tile of v_mfma_f32_16x16x4_f32: lane l holds D[(l >> 4) * 4 + r][l & 15], r = 0..3
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// The 16x16x4 matrix-core instruction of either precision (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64): lane l
// supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; the four accumulator registers of a lane hold column l & 15 of the
// rows row(l, r) - (l >> 4) * 4 + r in float, (l >> 4) + 4 r in double (scripts/microbench/mfma_f64_layout.hip, run on
// gfx950).
template <class S>
struct Mfma;
template <>
struct Mfma<float> {
  using acc = f32x4;
  using V2 = float2;
  using V4 = float4;
  static __device__ __forceinline__ acc mma(float a, float b, acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
  static __device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
};
template <>
struct Mfma<double> {
  using acc = f64x4;
  using V2 = double2;
  using V4 = double4;
  static __device__ __forceinline__ acc mma(double a, double b, acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
  static __device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
};

// Camera-major workgroups are launched as 8 * ceil(n_cams / 8) blocks and mapped so that one XCD
// (block b runs on XCD b % 8, each with its own L2) walks a CONTIGUOUS range of cameras: the records
// of a landmark sit next to each other in HBM and belong to cameras that are close in index, so the
// cache lines one camera's gather touches are shared with the cameras the same XCD works on at the
// same time instead of being fetched once per XCD.
__device__ __forceinline__ int xcd_swizzled_camera(int n_cams) {
  const int per = (n_cams + 7) / 8;
  return (blockIdx.x % 8) * per + blockIdx.x / 8;
}
inline int xcd_swizzled_grid(int n_cams) { return 8 * ((n_cams + 7) / 8); }

constexpr int kCamChunk = 32;

// pose_jacobian_scaling = 1 / (eps + sqrt(Jp_diag2))   (linearizor_qr.cpp:130-132)
template <class S>
__global__ void k_pose_scaling(const S* __restrict__ d2, S* __restrict__ sc, S eps, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sc[i] = S(1) / (eps + sqrt(d2[i]));
}

// ===========================================================================
// Implicit-Q  H*x  (same operator as k_hx up to rounding):
//   A^T A x = Jp^T W S^T S W^T Jp x,   W^T = (6 Givens) (H2 H1 H0),  S = drop Q1 rows
// evaluated with the FACTORS instead of the dense (2k x 9k) product: per block
// row the 9 Jacobian entries (JpS) and the 3 reflector entries (Vh), per landmark
// tau[3] and the 3x3 matrix Z (damp / drop / undamp of the top rows). Only
// orthogonal transforms are applied, never a normal-equation inverse.
// Mapping: ONE LANE PER BLOCK ROW, a landmark occupies an aligned group of P2
// lanes (P2 = 4..64 >= 2k), 64/P2 landmarks per wavefront; the six reflector dot
// products are segmented DPP reductions; results are transposed through LDS so
// that the scatter-add is 9 consecutive lanes <-> 9 consecutive floats.
// Traffic: (9 + 4) s + 4 bytes per block row instead of 9k s.
// ===========================================================================
template <class S, int P2>
__device__ __forceinline__ S seg_sum(S v) {
  // all lanes of each aligned P2-lane group receive the group sum
  v += dpp_mov0<0xb1>(v);  // quad_perm:[1,0,3,2]
  if (P2 >= 4) v += dpp_mov0<0x4e>(v);  // quad_perm:[2,3,0,1]
  if (P2 >= 8) v += dpp_mov0<0x141>(v);   // row_half_mirror
  if (P2 >= 16) v += dpp_mov0<0x140>(v);  // row_mirror
  if (P2 >= 32) v += __shfl_xor(v, 16);
  if (P2 >= 64) v += __shfl_xor(v, 32);
  return v;
}

struct ImplicitTiles {
  int tile_begin[6];  // first tile of class c (P2 = 4 << c); [5] = total
  int lm_begin[5];    // first landmark of class c
  int lm_end[5];
};

// wavefronts of k_s1_fused_obs (kernels_s1.hpp): pairs of row tiles, class by class; [5] = total
struct FusedObsWaves {
  int wave_begin[6];
};

template <class S, int P2>
__device__ __forceinline__ void hx_implicit_tile(const Params<S>& p, size_t T, int t_in_class,
                                                 int lm_begin, int lm_end, const S* __restrict__ x,
                                                 S* __restrict__ y, S* yb, int* cb, int lane,
                                                 int done, const S* __restrict__ dout, S* __restrict__ hx_u) {
  constexpr int LPW = 64 / P2;  // landmarks per wavefront (= per tile)
  const int seg = lane / P2, r = lane - P2 * seg;
  const int s = lm_begin + t_in_class * LPW + seg;
  const bool lm_ok = s < lm_end;
  // static lane maps (tile, lane) -> camera / block row; the Jacobian row (36 bytes) and the
  // reflector entries (16 bytes) are read straight from the stage-1 records: consecutive lanes
  // are consecutive rows, so a wave reads one contiguous 2.3 KB / 1 KB span
  int cam, row;
  tile_map(p, size_t(T), lane, cam, row);
  const bool act = cam >= 0;
  S jp[9];
  jp_row<S>(p.JpS, p.JpT, int64_t(act ? row : 0), jp);
  if (!act) {
#pragma unroll
    for (int c = 0; c < 9; ++c) jp[c] = S(0);
  }
  S v0 = S(0), v1 = S(0), v2 = S(0);
  if (act) {
    using V4 = typename std::conditional<sizeof(S) == 4, float4, double4>::type;
    const V4 vv = reinterpret_cast<const V4*>(p.Vh)[row];
    v0 = vv.x;
    v1 = vv.y;
    v2 = vv.z;
  }
  const S t0 = lm_ok ? p.tauH[3 * s + 0] : S(0), t1 = lm_ok ? p.tauH[3 * s + 1] : S(0),
          t2 = lm_ok ? p.tauH[3 * s + 2] : S(0);
  // u = Jp_row . x_cam: each row lane reads its camera's nine entries directly (the two rows of an
  // observation read the same 36 bytes; x is 9 n_c scalars and stays in L1/L2)
  const int obs_local = lane >> 1;
  if ((lane & 1) == 0) cb[obs_local] = cam;
  S u = S(0);
  {
    const S* __restrict__ xc = x + 9 * (act ? cam : 0);
#pragma unroll
    for (int c = 0; c < 9; ++c) u += jp[c] * xc[c];
  }
  // W^T: reflectors 0,1,2
  u -= t0 * seg_sum<S, P2>(v0 * u) * v0;
  u -= t1 * seg_sum<S, P2>(v1 * u) * v1;
  u -= t2 * seg_sum<S, P2>(v2 * u) * v2;
  // top three rows: damp, drop Q1, undamp == 3x3 matrix Z
  {
    const int base = lane - r;
    const S u0 = __shfl(u, base), u1 = __shfl(u, base + 1), u2 = __shfl(u, base + 2);
    if (act && r < 3) {
      const S* Z = p.Zd + 9 * s + 3 * r;
      u = Z[0] * u0 + Z[1] * u1 + Z[2] * u2;
    }
  }
  // W: reflectors 2,1,0
  u -= t2 * seg_sum<S, P2>(v2 * u) * v2;
  u -= t1 * seg_sum<S, P2>(v1 * u) * v1;
  u -= t0 * seg_sum<S, P2>(v0 * u) * v0;
  if (hx_u) {
    // deterministic form (k_hx_det_gather below): the row's entry of (I - Q1 Q1^T)-projected J x, summed camera-major
    if (act) hx_u[row] = u;
    return;
  }
  // y_obs = Jp_obs^T u_obs: add the two rows of an observation (lanes r, r^1),
  // transpose through LDS so that 9 consecutive lanes hold one observation
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    S v = jp[c] * u;
    v += dpp_mov0<0xb1>(v);  // partner row of the same observation
    if ((lane & 1) == 0) yb[9 * obs_local + c] = act ? v : S(0);
  }
  wave_lds_fence();
  // (measured: this scatter is ~100 of the kernel's ~265 us on venice - about 7.5 M
  //  36-byte requests; pre-aggregating per workgroup after sorting landmarks by camera
  //  made it worse because same-address atomics serialise; accumulating into a
  //  workgroup-private 64 KB LDS copy of y with ds_add_f32 from persistent 1024-thread
  //  workgroups, with or without register prefetch of the next tile, also measured
  //  slower: 318 us. A flat atomic on an LDS pointer faults on gfx950, by the way.)
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int e = q * 64 + lane;
    if (e < 288) {
      const int ol = e / 9, c = e - 9 * ol;
      const int cc = cb[ol];
      if (cc >= 0 && !done) atomic_add(y + 9 * cc + c, dout ? yb[e] * dout[9 * cc + c] : yb[e]);
    }
  }
}

// all tiled classes (k <= 32) in ONE launch: wave -> tile -> class
template <class S>
__global__ __launch_bounds__(256) void k_hx_implicit(Params<S> p, ImplicitTiles it,
                                                     const S* __restrict__ x, S* __restrict__ y,
                                                     const S* __restrict__ dout,
                                                     const int* __restrict__ done_flag, S* __restrict__ hx_u) {
  __shared__ S ybuf[4][32 * 9 + 8];
  __shared__ int cbuf[4][32];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int T = blockIdx.x * 4 + wave;
  if (T >= it.tile_begin[5]) return;  // whole wave idle
  // the PCG "done" flag is fetched alongside the data and only gates the scatter, so
  // it does not add a memory round trip in front of the loads
  const int done = done_flag ? *done_flag : 0;
  S* yb = ybuf[wave];
  int* cb = cbuf[wave];
  if (T >= it.tile_begin[4])
    hx_implicit_tile<S, 64>(p, T, T - it.tile_begin[4], it.lm_begin[4], it.lm_end[4], x, y, yb, cb, lane, done, dout, hx_u);
  else if (T >= it.tile_begin[3])
    hx_implicit_tile<S, 32>(p, T, T - it.tile_begin[3], it.lm_begin[3], it.lm_end[3], x, y, yb, cb, lane, done, dout, hx_u);
  else if (T >= it.tile_begin[2])
    hx_implicit_tile<S, 16>(p, T, T - it.tile_begin[2], it.lm_begin[2], it.lm_end[2], x, y, yb, cb, lane, done, dout, hx_u);
  else if (T >= it.tile_begin[1])
    hx_implicit_tile<S, 8>(p, T, T - it.tile_begin[1], it.lm_begin[1], it.lm_end[1], x, y, yb, cb, lane, done, dout, hx_u);
  else
    hx_implicit_tile<S, 4>(p, T, T - it.tile_begin[0], it.lm_begin[0], it.lm_end[0], x, y, yb, cb, lane, done, dout, hx_u);
}

// ---------------------------------------------------------------------------
// Tile loads of the persistent, software-pipelined form of the product (k_hx_implicit_lds below):
//   * every load of a tile is unconditional (clamped indices, zeros selected afterwards), so the nine
//     entries of a row are two 16-byte loads + one 4-byte load, x likewise (the one-tile-per-wave form
//     above issues nine exec-masked 4-byte loads);
//   * the per-landmark scalars (tau, the Z row) travel with the records instead of being fetched in
//     the middle of the reflector chain;
//   * a wave walks the tiles T, T + W, T + 2W, ...: the lane maps of tile i+2 and the records of tile
//     i+1 are in flight while tile i is reduced (registers ping-pong between two sets; the loop is
//     unrolled by two and no load is conditional, so no load result is ever copied - a copy would wait
//     for its load right after issuing it).
// (Measured on venice, float: the same pipeline with device-scope atomics instead of the LDS copy runs at
//  252 us against 238 us for the one-tile-per-wave form - a wave's loads queue behind its own outstanding
//  atomics in the in-order vmcnt counter - so that variant is not kept.)
// ---------------------------------------------------------------------------
template <class S>
struct HxTileData {
  S jp[9], xc[9], v0, v1, v2, t0, t1, t2, z0, z1, z2;
};

__device__ __forceinline__ int hx_tile_class(const ImplicitTiles& it, int T) {
  return int(T >= it.tile_begin[1]) + int(T >= it.tile_begin[2]) + int(T >= it.tile_begin[3]) +
         int(T >= it.tile_begin[4]);
}

template <class S>
__device__ __forceinline__ void hx_tile_load(const Params<S>& p, const ImplicitTiles& it, int T, int cam, int row,
                                             int lane, const S* __restrict__ x, HxTileData<S>& d) {
  using V4 = typename std::conditional<sizeof(S) == 4, float4, double4>::type;
  const int cls = hx_tile_class(it, T);  // wave-uniform
  const int sh = 2 + cls;                // P2 = 1 << sh lanes per landmark
  const int seg = lane >> sh, r = lane & ((1 << sh) - 1);
  const int lb = cls == 0 ? it.lm_begin[0] : cls == 1 ? it.lm_begin[1] : cls == 2 ? it.lm_begin[2]
               : cls == 3 ? it.lm_begin[3] : it.lm_begin[4];
  const int le = cls == 0 ? it.lm_end[0] : cls == 1 ? it.lm_end[1] : cls == 2 ? it.lm_end[2]
               : cls == 3 ? it.lm_end[3] : it.lm_end[4];
  const int tb = cls == 0 ? it.tile_begin[0] : cls == 1 ? it.tile_begin[1] : cls == 2 ? it.tile_begin[2]
               : cls == 3 ? it.tile_begin[3] : it.tile_begin[4];
  const int s = min(lb + ((T - tb) << (6 - sh)) + seg, le - 1);  // clamped: padding segments read a valid landmark
  const int64_t rw = cam >= 0 ? row : 0;
  const int cc = cam >= 0 ? cam : 0;
  jp_row<S>(p.JpS, p.JpT, rw, d.jp);
  const V4 vv = reinterpret_cast<const V4*>(p.Vh)[rw];
  d.v0 = vv.x;
  d.v1 = vv.y;
  d.v2 = vv.z;
  // (measured in round 6 and not kept: each lane of an observation's pair gathering five of the camera's nine entries
  //  and a DPP exchange of the halves - 127.9 / 128.6 us against 126.1 / 126.8 on one box: the product is not bound by
  //  its gather instructions)
  const S* __restrict__ xc = x + 9 * cc;
#pragma unroll
  for (int c = 0; c < 9; ++c) d.xc[c] = xc[c];
  const S* __restrict__ th = p.tauH + 3 * int64_t(s);
  d.t0 = th[0];
  d.t1 = th[1];
  d.t2 = th[2];
  const S* __restrict__ Z = p.Zd + 9 * int64_t(s) + 3 * min(r, 2);
  d.z0 = Z[0];
  d.z1 = Z[1];
  d.z2 = Z[2];
}

// ---------------------------------------------------------------------------
// The product with a WORKGROUP-PRIVATE copy of y in LDS (used when 9 n_c scalars fit: venice-1778 in
// float is 64 008 bytes of the CU's 160 KB). The device-scope float atomics of the forms above are
// fabric transactions (PMC: 0.22 GB of write traffic per product for 0.18 GB of payload, and a wave's
// loads queue behind its own outstanding atomics in the in-order vmcnt counter); here a tile's
// contributions are ds_add_f64's into a DOUBLE LDS copy (measured on gfx950, scripts/microbench/lds_atomics.hip:
// a 64-lane ds_add_f32 with random addresses takes ~80 ns of the CU's LDS, ds_add_f64 ~10 ns, ds_add_u32 ~7 ns -
// with float accumulators this kernel ran at 265 us, slower than the global atomics) - outside the vector-memory queue - and
// each of the (one per CU) persistent 1024-thread workgroups flushes its copy once at the end with fully
// coalesced atomics: 256 x 9 n_c instead of 9 n_obs. The two rows of an observation share the nine adds
// (row 2i takes components 0,2,4,6,8, row 2i+1 takes 1,3,5,7).
// ---------------------------------------------------------------------------
// ALL: the window holds every camera (venice-1778 and smaller) - no test, no path of global atomics in the tile loop
template <class S, int P2, bool ALL>
__device__ __forceinline__ void hx_tile_compute_lds(const HxTileData<S>& d, int cam, int lane, double* ylds,
                                                    int cam_lo, int win, S* __restrict__ y,
                                                    const S* __restrict__ dout) {
  const int r = lane & (P2 - 1);
  const bool act = cam >= 0;
  S u = S(0);
#pragma unroll
  for (int c = 0; c < 9; ++c) u += d.jp[c] * d.xc[c];
  u = act ? u : S(0);
  const S v0 = act ? d.v0 : S(0), v1 = act ? d.v1 : S(0), v2 = act ? d.v2 : S(0);
  u -= d.t0 * seg_sum<S, P2>(v0 * u) * v0;
  u -= d.t1 * seg_sum<S, P2>(v1 * u) * v1;
  u -= d.t2 * seg_sum<S, P2>(v2 * u) * v2;
  {
    const int base = lane - r;
    const S u0 = __shfl(u, base), u1 = __shfl(u, base + 1), u2 = __shfl(u, base + 2);
    if (act && r < 3) u = d.z0 * u0 + d.z1 * u1 + d.z2 * u2;
  }
  u -= d.t2 * seg_sum<S, P2>(v2 * u) * v2;
  u -= d.t1 * seg_sum<S, P2>(v1 * u) * v1;
  u -= d.t0 * seg_sum<S, P2>(v0 * u) * v0;
  const int par = lane & 1;
  // cameras outside the workgroup's window [cam_lo, cam_lo + win) go straight to y (rare: wrap-around
  // and long tracks when the landmarks are sorted by camera; never when the window holds every camera)
  const int rel = (act ? cam : cam_lo) - cam_lo;
  const bool inside = ALL || unsigned(rel) < unsigned(win);
  double* yc = ylds + 9 * (inside ? rel : 0) + par;
  S* yg = y + 9 * (act ? cam : 0) + par;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    // both rows of the observation receive the pair sums of components 2q and 2q+1; each adds one
    const S ve = d.jp[2 * q] * u;
    const S se = ve + dpp_mov0<0xb1>(ve);
    S mine = se;
    if (q < 4) {
      const S vo = d.jp[2 * q + 1] * u;
      const S so = vo + dpp_mov0<0xb1>(vo);
      mine = par ? so : se;
    }
    if (act && (q < 4 || par == 0)) {
      if (inside)
        lds_atomic_add(yc + 2 * q, double(mine));
      else
        atomic_add(yg + 2 * q, dout ? mine * dout[9 * cam + par + 2 * q] : mine);
    }
  }
}

// One persistent workgroup's share. With the landmarks sorted by first camera inside each track-length
// class, the tiles form a few RUNS of ascending first camera; workgroup g takes, from every run, the tiles
// whose first camera lies in its camera range, so everything it adds lands in a short camera interval
// starting at `cam_lo` (tracks are local; wrap-around and very long tracks are the exception and go to y
// directly). When all cameras fit the LDS window the ranges simply split the tiles evenly.
constexpr int kHxMaxRuns = 16;
struct HxChunk {
  int cam_lo, n_ranges;
  int tile_begin[kHxMaxRuns], tile_end[kHxMaxRuns];
};

// v-th tile of a chunk (v clamped to the chunk's last tile)
__device__ __forceinline__ int hx_chunk_tile(const HxChunk& ch, int v, int n_total) {
  v = min(v, n_total - 1);
  int T = ch.tile_begin[0];
  for (int r = 0; r < ch.n_ranges; ++r) {
    const int len = ch.tile_end[r] - ch.tile_begin[r];
    if (v < len) {
      T = ch.tile_begin[r] + v;
      break;
    }
    v -= len;
  }
  return T;
}

// NT threads per workgroup: 1024 caps a lane at 128 VGPRs - enough for float (114), 104 bytes of scratch per lane
// for double; the 512-thread instance has 256 and spills nothing, at half the waves per CU (RBA_HX_THREADS).
// (defined below, with the kernel of its own that the non-persistent product uses)
template <class S, int RCH>
__device__ __forceinline__ void hx_wide_landmark(const Params<S>& p, int s, const S* __restrict__ x, S* __restrict__ y,
                                                 const S* __restrict__ dout, S* yb, int* cb, int lane,
                                                 S* __restrict__ hx_u = nullptr);
constexpr int kHxWideScalars = 32 * 9 + 8;  // LDS scalars of a wavefront in hx_wide_landmark (+ 32 ints)

// range of the landmarks with 32 < k <= 64 (RCH = 2)
struct HxWideRanges {
  int begin[1], end[1];
};

template <class S, int NT, bool ALL>
__global__ __launch_bounds__(NT) void k_hx_implicit_lds(Params<S> p, ImplicitTiles it,
                                                        const HxChunk* __restrict__ chunks, int win,
                                                        const S* __restrict__ x, S* __restrict__ y,
                                                        const S* __restrict__ dout,
                                                        const int* __restrict__ done_flag, HxWideRanges wide) {
  // `dout` (compact stage 2: the Jacobian rows are unscaled, x arrives pre-multiplied by the pose scaling D):
  // the result is multiplied by D where it leaves the workgroup; nullptr = rows already scaled
  extern __shared__ __align__(16) unsigned char hx_lds_raw[];
  __shared__ HxChunk ch;
  double* ylds = reinterpret_cast<double*>(hx_lds_raw);
  if (done_flag && *done_flag) return;  // uniform over the grid: nothing is added after the PCG has terminated
  if (threadIdx.x < sizeof(HxChunk) / sizeof(int))
    reinterpret_cast<int*>(&ch)[threadIdx.x] = reinterpret_cast<const int*>(chunks + blockIdx.x)[threadIdx.x];
  const int nwin = 9 * win;
  for (int i = threadIdx.x; i < nwin; i += NT) ylds[i] = 0.0;
  __syncthreads();
  const int cam_lo = ch.cam_lo;
  int nV = 0;
  for (int r = 0; r < ch.n_ranges; ++r) nV += ch.tile_end[r] - ch.tile_begin[r];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int W = NT / 64;
  int vA = wave;
  if (vA < nV) {
    int vB = vA + W;
    int TA = hx_chunk_tile(ch, vA, nV), TB = hx_chunk_tile(ch, vB, nV);
    int camA, rowA, camB, rowB;
    tile_map(p, size_t(TA), lane, camA, rowA);
    tile_map(p, size_t(TB), lane, camB, rowB);
    HxTileData<S> dA, dB;
    hx_tile_load(p, it, TA, camA, rowA, lane, x, dA);
    auto compute = [&](int T, const HxTileData<S>& d, int cam) {
      switch (hx_tile_class(it, T)) {
        case 0: hx_tile_compute_lds<S, 4, ALL>(d, cam, lane, ylds, cam_lo, win, y, dout); break;
        case 1: hx_tile_compute_lds<S, 8, ALL>(d, cam, lane, ylds, cam_lo, win, y, dout); break;
        case 2: hx_tile_compute_lds<S, 16, ALL>(d, cam, lane, ylds, cam_lo, win, y, dout); break;
        case 3: hx_tile_compute_lds<S, 32, ALL>(d, cam, lane, ylds, cam_lo, win, y, dout); break;
        default: hx_tile_compute_lds<S, 64, ALL>(d, cam, lane, ylds, cam_lo, win, y, dout); break;
      }
    };
    for (;;) {
      const int vC = vB + W;
      const int TC = hx_chunk_tile(ch, vC, nV);
      int camC, rowC;
      tile_map(p, size_t(TC), lane, camC, rowC);
      hx_tile_load(p, it, TB, camB, rowB, lane, x, dB);
      compute(TA, dA, camA);
      if (vB >= nV) break;
      const int vD = vC + W;
      const int TD = hx_chunk_tile(ch, vD, nV);
      int camD, rowD;
      tile_map(p, size_t(TD), lane, camD, rowD);
      hx_tile_load(p, it, TC, camC, rowC, lane, x, dA);
      compute(TB, dB, camB);
      if (vC >= nV) break;
      TA = TC;
      camA = camC;
      vB = vD;
      TB = TD;
      camB = camD;
      rowB = rowD;
    }
  }
  __syncthreads();
  // flush (zeros are skipped: with camera-sorted landmarks a workgroup touches a small part of its window);
  // every workgroup starts at its own offset so that copies of the same entries do not arrive together
  const int start = int((int64_t(blockIdx.x) * nwin) / gridDim.x);
  S* __restrict__ yw = y + 9 * cam_lo;
  const S* __restrict__ dw = dout ? dout + 9 * cam_lo : nullptr;
  for (int i = threadIdx.x; i < nwin; i += NT) {
    int j = i + start;
    j = j >= nwin ? j - nwin : j;
    const double v = ylds[j];
    if (v != 0.0) atomic_add(yw + j, dw ? S(v * double(dw[j])) : S(v));
  }
  // The landmarks with 32 < k <= 64 observations (a wavefront each, straight into y with atomics): a few hundred on
  // venice-1778 - as a kernel of their own they cost a launch of 8 us per product, mostly latency; here the wavefronts
  // that are done with their tiles take them (64 < k <= 112 keep their kernel: four rows per lane do not fit the
  // 128 registers of this one). Their LDS buffers reuse the window (the launch reserves at least
  // W * (kHxWideScalars * sizeof(S) + 128) bytes).
  if (wide.end[0] > wide.begin[0]) {
    __syncthreads();  // the flush has read the window
    S* yb = reinterpret_cast<S*>(hx_lds_raw) + wave * kHxWideScalars;
    int* cb = reinterpret_cast<int*>(hx_lds_raw + size_t(W) * kHxWideScalars * sizeof(S)) + wave * 32;
    const int stride = int(gridDim.x) * W;
    for (int s = wide.begin[0] + int(blockIdx.x) * W + wave; s < wide.end[0]; s += stride)
      hx_wide_landmark<S, 2>(p, s, x, y, dout, yb, cb, lane);
  }
}

// rows of one landmark spread over RCH x 64 lanes (32 < k <= 112): one landmark
// per wavefront, reflector sums via wave_sum
// One landmark with 32 < k <= 32 RCH observations on one wavefront (rows rc * 64 + lane): H x contribution added to y
// with atomics; `yb` (32 * 9 + 8 scalars) and `cb` (32 ints) are LDS buffers of this wavefront.
template <class S, int RCH>
__device__ __forceinline__ void hx_wide_landmark(const Params<S>& p, int s, const S* __restrict__ x, S* __restrict__ y,
                                                 const S* __restrict__ dout, S* yb, int* cb, int lane,
                                                 S* __restrict__ hx_u) {
  const int k = p.lm_k[s];
  const int64_t o0 = p.lm_obs[s];
  S jp[RCH][9], u[RCH], v[3][RCH];
  int cam[RCH];
  bool act[RCH];
#pragma unroll
  for (int rc = 0; rc < RCH; ++rc) {
    const int r = rc * 64 + lane;
    act[rc] = r < 2 * k;
    cam[rc] = 0;
    u[rc] = S(0);
    v[0][rc] = v[1][rc] = v[2][rc] = S(0);
#pragma unroll
    for (int c = 0; c < 9; ++c) jp[rc][c] = S(0);
    if (act[rc]) {
      const int64_t obs = o0 + (r >> 1);
      cam[rc] = p.obs_cam[obs];
      jp_row<S>(p.JpS, p.JpT, 2 * obs + (r & 1), jp[rc]);
      const S* vh = p.Vh + 4 * (2 * o0 + r);
      v[0][rc] = vh[0];
      v[1][rc] = vh[1];
      v[2][rc] = vh[2];
      S acc = S(0);
#pragma unroll
      for (int c = 0; c < 9; ++c) acc += jp[rc][c] * x[9 * cam[rc] + c];
      u[rc] = acc;
    }
  }
  const S tau[3] = {p.tauH[3 * s], p.tauH[3 * s + 1], p.tauH[3 * s + 2]};
  auto reflect = [&](int m) {
    S d = S(0);
#pragma unroll
    for (int rc = 0; rc < RCH; ++rc) d += v[m][rc] * u[rc];
    d = tau[m] * wave_sum(d);
#pragma unroll
    for (int rc = 0; rc < RCH; ++rc) u[rc] -= d * v[m][rc];
  };
  reflect(0);
  reflect(1);
  reflect(2);
  {
    const S u0 = read_lane(u[0], 0), u1 = read_lane(u[0], 1), u2 = read_lane(u[0], 2);
    if (lane < 3) {
      const S* Z = p.Zd + 9 * s + 3 * lane;
      u[0] = Z[0] * u0 + Z[1] * u1 + Z[2] * u2;
    }
  }
  reflect(2);
  reflect(1);
  reflect(0);
  if (hx_u) {  // deterministic form: see k_hx_det_gather
#pragma unroll
    for (int rc = 0; rc < RCH; ++rc)
      if (act[rc]) hx_u[2 * o0 + rc * 64 + lane] = u[rc];
    return;
  }
  const int obs_local = lane >> 1;
#pragma unroll
  for (int rc = 0; rc < RCH; ++rc) {
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      S w = jp[rc][c] * u[rc];
      w += dpp_mov0<0xb1>(w);
      if ((lane & 1) == 0) yb[9 * obs_local + c] = act[rc] ? w : S(0);
    }
    if ((lane & 1) == 0) cb[obs_local] = act[rc] ? cam[rc] : -1;
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int e = q * 64 + lane;
      if (e < 288) {
        const int ol = e / 9, c = e - 9 * ol;
        const int cc = cb[ol];
        if (cc >= 0) atomic_add(y + 9 * cc + c, dout ? yb[e] * dout[9 * cc + c] : yb[e]);
      }
    }
    wave_lds_fence();
  }
}

template <class S, int RCH>
__global__ __launch_bounds__(256) void k_hx_implicit_wide(Params<S> p, int lm_begin, int lm_end,
                                                          const S* __restrict__ x,
                                                          S* __restrict__ y, const S* __restrict__ dout,
                                                          const int* __restrict__ done_flag, S* __restrict__ hx_u) {
  __shared__ S ybuf[4][kHxWideScalars];
  __shared__ int cbuf[4][32];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s = lm_begin + blockIdx.x * 4 + wave;
  if (s >= lm_end) return;
  if (done_flag && *done_flag) return;
  hx_wide_landmark<S, RCH>(p, s, x, y, dout, ybuf[wave], cbuf[wave], lane, hx_u);
}

// ---------------------------------------------------------------------------
// The DETERMINISTIC form of the matrix-free product (RBA_DETERMINISTIC=1). The forms above
// add into y in whatever order the hardware serves their atomics (ds_add_f64 into the workgroup's window, then float
// atomics into y): two runs of the same float32 solve differ in the last bits of every product, and a 300-iteration
// PCG amplifies that into +- 1 % of its iteration count. Ordering the flush of the windows alone would not do - the
// additions INTO a window race as well. Here the landmark-major kernels stop after the reflector chain and store the
// row entries u = P J x (one scalar per block row, `hx_u`), and this kernel - one workgroup per camera over the
// camera's observation list, the CSC index of the camera-major passes - sums y_c = D_c sum_o Jp_o^T u_o in double in a
// fixed order. Costs a second pass over the Jacobian rows (+ 80 B per observation): about twice the product's time;
// it is a mode for reproducing a run bit by bit, not the default.
// ---------------------------------------------------------------------------
template <class S>
__global__ __launch_bounds__(256) void k_hx_det_gather(Params<S> p, const S* __restrict__ hx_u, S* __restrict__ y,
                                                       const S* __restrict__ dout,
                                                       const int* __restrict__ done_flag) {
  __shared__ double sm[4][9];
  const int c = xcd_swizzled_camera(p.n_cams);
  if (c >= p.n_cams) return;
  if (done_flag && *done_flag) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t t0 = p.cam_obs_off[c], t1 = p.cam_obs_off[c + 1];
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t t = t0 + tid; t < t1; t += 256) {
    const int64_t o = p.cam_obs[t];
    const double u0 = double(hx_u[2 * o]), u1 = double(hx_u[2 * o + 1]);
    S jp[18];
    jp_obs<S>(p.JpS, p.JpT, o, jp);
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = fma(double(jp[9 + j]), u1, fma(double(jp[j]), u0, acc[j]));
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) acc[j] = wave_sum(acc[j]);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 9; ++j) sm[wave][j] = acc[j];
  }
  __syncthreads();
  if (tid < 9) {
    const double t = (sm[0][tid] + sm[1][tid]) + (sm[2][tid] + sm[3][tid]);
    const int i = 9 * c + tid;
    y[i] += dout ? S(t * double(dout[i])) : S(t);
  }
}

// ===========================================================================
// E0 * v = sum_l (Q1^T Jp)_l^T (Q1^T Jp)_l v_l  with the DAMPED top rows, i.e.
// Jp^T Jl (Jl^T Jl + lambda I)^-1 Jl^T Jp v of the power-series (PoBA)
// preconditioner (right_mul_e0, src/rootba/cg/preconditioner.hpp:223-245) —
// expressed through the square-root factors that stage 2 already holds, so one
// application reads 27 scalars per observation instead of the sparse Jacobians
// plus a 3x3 inverse per landmark.
// ===========================================================================
template <class S, int CH>
__global__ __launch_bounds__(256) void k_e0(Params<S> p, int lm_begin, int lm_end,
                                            const S* __restrict__ v, S* __restrict__ y,
                                            const int* __restrict__ done_flag, S* __restrict__ e0_w) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s = lm_begin + blockIdx.x * 4 + wave;
  if (s >= lm_end) return;
  if (done_flag && *done_flag) return;
  const int k = p.lm_k[s];
  const int64_t o0 = p.lm_obs[s];
  const S* __restrict__ Td = p.topd + kTd * o0;
  const int lane9 = lane / 9, comp = lane - 9 * lane9;
  S t[3][CH];
  int yidx[CH];
  bool act[CH];
  S w[3] = {S(0), S(0), S(0)};
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) {
    const int islot = 7 * ch + lane9;
    act[ch] = lane < 63 && islot < k;
    const int cam = act[ch] ? p.obs_cam[o0 + islot] : 0;
    yidx[ch] = 9 * cam + comp;
    const S xv = act[ch] ? v[yidx[ch]] : S(0);
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      t[m][ch] = act[ch] ? Td[kTd * islot + 9 * m + comp] : S(0);
      w[m] += t[m][ch] * xv;
    }
  }
#pragma unroll
  for (int m = 0; m < 3; ++m) w[m] = wave_sum(w[m]);
  if (e0_w) {  // deterministic form: the landmark's three sums, applied camera-major by k_e0_det_gather
    if (lane < 3) e0_w[3 * size_t(s) + lane] = lane == 0 ? w[0] : (lane == 1 ? w[1] : w[2]);
    return;
  }
#pragma unroll
  for (int ch = 0; ch < CH; ++ch)
    if (act[ch]) atomic_add(y + yidx[ch], t[0][ch] * w[0] + t[1][ch] * w[1] + t[2][ch] * w[2]);
}

// deterministic E0 v (RBA_DETERMINISTIC=1, see k_hx_det_gather): y_c += sum_o topd_o^T w_{l(o)} over the camera's
// observation list, double, fixed order
template <class S>
__global__ __launch_bounds__(256) void k_e0_det_gather(Params<S> p, const S* __restrict__ e0_w, S* __restrict__ y,
                                                       const int* __restrict__ done_flag) {
  __shared__ double sm[4][9];
  const int c = xcd_swizzled_camera(p.n_cams);
  if (c >= p.n_cams) return;
  if (done_flag && *done_flag) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t t0 = p.cam_obs_off[c], t1 = p.cam_obs_off[c + 1];
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t t = t0 + tid; t < t1; t += 256) {
    const int64_t o = p.cam_obs[t];
    const S* __restrict__ w = e0_w + 3 * size_t(p.obs_lm[o]);
    const double w0 = double(w[0]), w1 = double(w[1]), w2 = double(w[2]);
    const S* __restrict__ td = p.topd + kTd * o;
#pragma unroll
    for (int j = 0; j < 9; ++j)
      acc[j] = fma(double(td[18 + j]), w2, fma(double(td[9 + j]), w1, fma(double(td[j]), w0, acc[j])));
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) acc[j] = wave_sum(acc[j]);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 9; ++j) sm[wave][j] = acc[j];
  }
  __syncthreads();
  if (tid < 9) y[9 * c + tid] += S((sm[0][tid] + sm[1][tid]) + (sm[2][tid] + sm[3][tid]));
}

// ===========================================================================
// Back-substitution (back_substitute ipp:212-284; loop linearization_qr.hpp:165-179)
//   delta = -Rd^{-1} (Q1^T r + Q1^T Jp x)        on the DAMPED top rows
//   l_diff -= g^T (g/2 + Q^T r), g = (Q^T J)[x; delta] on the UNDAMPED 2k rows
//   p_w += delta o Jl_col_scale
// Q is orthogonal, so  g^T g = |J inc|^2  and  g^T Q^T r = (J inc)^T r  with
// J inc = Jp x + Jl delta per observation: the model-cost term is evaluated from the
// 26-scalar observation records of stage 1 instead of streaming the dense 2k x 9k block
// (1.1 GB instead of 3.9 GB on venice), delta from the damped top rows as before.
//   k_bs_obs      one thread per observation:  topd_o x (3), Jp_o x (2)
//   k_bs_landmark one thread per landmark (fixed order): delta, l_diff term, p_w update
// ===========================================================================
template <class S>
__global__ __launch_bounds__(256) void k_bs_obs(Params<S> p, const S* __restrict__ x, int64_t o_begin,
                                                int64_t n_obs) {
  const int64_t o = o_begin + int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (o >= n_obs) return;
  const S* __restrict__ xc = x + 9 * p.obs_cam[o];
  S jp[18];
  jp_obs<S>(p.JpS, p.JpT, o, jp);
  S xv[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) xv[c] = xc[c];
  S out[5] = {S(0), S(0), S(0), S(0), S(0)};
  // x arrives pre-multiplied by the pose scaling; topd x = W' (Jp D x)
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    out[3] += jp[c] * xv[c];
    out[4] += jp[9 + c] * xv[c];
  }
  const S* __restrict__ w = p.W8 + 8 * (o - p.w8_begin);
  out[0] = w[0] * out[3] + w[1] * out[4];
  out[1] = w[2] * out[3] + w[3] * out[4];
  out[2] = w[4] * out[3] + w[5] * out[4];
#pragma unroll
  for (int m = 0; m < 5; ++m) p.bsO[5 * o + m] = out[m];
}

template <class S>
__global__ __launch_bounds__(256) void k_bs_landmark(Params<S> p, int lm_begin, int lm_end) {
  const int s = lm_begin + blockIdx.x * 256 + threadIdx.x;
  if (s >= lm_end) return;
  const int64_t ob = p.lm_obs[s], oe = p.lm_obs[s + 1];
  S rhs[3] = {p.q1trd[3 * s], p.q1trd[3 * s + 1], p.q1trd[3 * s + 2]};
  for (int64_t o = ob; o < oe; ++o) {
    rhs[0] += p.bsO[5 * o];
    rhs[1] += p.bsO[5 * o + 1];
    rhs[2] += p.bsO[5 * o + 2];
  }
  const S* Rd = p.Rd + 6 * s;
  S inc[3];
  inc[2] = rhs[2] / Rd[5];
  inc[1] = (rhs[1] - Rd[4] * inc[2]) / Rd[3];
  inc[0] = (rhs[0] - Rd[1] * inc[1] - Rd[2] * inc[2]) / Rd[0];
#pragma unroll
  for (int m = 0; m < 3; ++m) inc[m] = -inc[m];
  S acc = S(0);
  for (int64_t o = ob; o < oe; ++o) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const S* jl = p.JlS + 6 * o + 3 * r;
      const S v = p.bsO[5 * o + 3 + r] + jl[0] * inc[0] + jl[1] * inc[1] + jl[2] * inc[2];
      acc += v * (S(0.5) * v + p.rS[2 * o + r]);
    }
  }
  finish_landmark(p, s, inc, acc);
}

// the same, one WAVEFRONT per landmark (lanes run over the observations): for the few landmarks with 32 < k <= 112
// of the staged configuration, where a thread per landmark is a handful of workgroups walking ~100 rows serially
template <class S>
__global__ __launch_bounds__(256) void k_bs_landmark_wave(Params<S> p, int lm_begin, int lm_end) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s = lm_begin + blockIdx.x * 4 + wave;
  if (s >= lm_end) return;
  const int64_t ob = p.lm_obs[s], oe = p.lm_obs[s + 1];
  S r0 = S(0), r1 = S(0), r2 = S(0);
  for (int64_t o = ob + lane; o < oe; o += 64) {
    r0 += p.bsO[5 * o];
    r1 += p.bsO[5 * o + 1];
    r2 += p.bsO[5 * o + 2];
  }
  S rhs[3] = {p.q1trd[3 * s] + wave_sum(r0), p.q1trd[3 * s + 1] + wave_sum(r1), p.q1trd[3 * s + 2] + wave_sum(r2)};
  const S* Rd = p.Rd + 6 * s;
  S inc[3];
  inc[2] = rhs[2] / Rd[5];
  inc[1] = (rhs[1] - Rd[4] * inc[2]) / Rd[3];
  inc[0] = (rhs[0] - Rd[1] * inc[1] - Rd[2] * inc[2]) / Rd[0];
#pragma unroll
  for (int m = 0; m < 3; ++m) inc[m] = -inc[m];
  S acc = S(0);
  for (int64_t o = ob + lane; o < oe; o += 64) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const S* jl = p.JlS + 6 * o + 3 * r;
      const S v = p.bsO[5 * o + 3 + r] + jl[0] * inc[0] + jl[1] * inc[1] + jl[2] * inc[2];
      acc += v * (S(0.5) * v + p.rS[2 * o + r]);
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    finish_landmark(p, s, inc, acc);
  }
}

// The same back-substitution for the tiled landmarks (k <= 32) in ONE pass, lane per block row
// (the wave tiles of k_hx_implicit): u = Jp x per row; Q^T u by the three reflectors (segmented
// reductions); its three top entries rotated by the landmark's damping rotations are exactly
// topd x, so the stored damped top rows are not read at all; delta from the damped triangle; the
// model-cost term sum v (v / 2 + r) of the undamped rows v = u + Jl delta is evaluated in the ROTATED frame - Q is
// orthogonal, Q^T v = Q^T u + [R0 delta; 0] and Q^T r are at hand (the reflected u and the fourth entry of the
// reflector record) - so neither the pre-QR Jl rows nor the residuals are read (round 2 read them: 152 bytes per
// observation; now 120: Jacobian rows, reflector records, the two lane maps), and the QR pass of the wave tiles no
// longer writes them.
//
// One tile per wave (the persistent, software-pipelined form of the product - k_hx_implicit_lds - was measured here too:
// 239 us against 219 on venice-1778; without atomics in the way the hardware's eight waves per SIMD hide the latency
// better than four waves with a prefetched tile), every load unconditional (clamped indices, zeros selected afterwards).
// The landmark's 30 scalars (rotations, damped triangle, damped Q1^T r, tau, undamped triangle) are fetched
// CO-OPERATIVELY: lane q of a quad loads piece q of the rotation record and element q (and 4 + q) of the short arrays,
// and quad-broadcast DPP moves hand them round - every quad of the landmark's lane group ends up with everything -
// instead of 27 loads of one scalar per lane that all lanes of the group repeat (measured on venice-1778 with those
// loads stubbed out: 34 us of the pass's 258). (One 128-byte record per landmark written by stage 2 was measured as
// well: the back-substitution gained the same, k_s2_obs lost 45 us on the eight sparse 16-byte stores per landmark.)
template <class S>
struct BsTileData {
  using V4 = typename std::conditional<sizeof(S) == 4, float4, double4>::type;
  S jp[9], xc[9];
  V4 vv;  // v0 v1 v2 (Q^T r)[row]
  V4 g;   // piece (lane & 3) of the landmark's rotation record [c 6 | s 6 | ...]
  S rd0, rd1, q1, tau, r00, r01;  // elements q / 4 + q (clamped) of Rd, q of Q1^T r and tau, q / 4 + q of R0
};

template <class S>
__device__ __forceinline__ void bs_tile_load(const Params<S>& p, const ImplicitTiles& it, int T, int cam, int row,
                                             int lane, const S* __restrict__ x, BsTileData<S>& d) {
  using V4 = typename BsTileData<S>::V4;
  const int cls = hx_tile_class(it, T);  // wave-uniform
  const int sh = 2 + cls;                // P2 = 1 << sh lanes per landmark
  const int seg = lane >> sh;
  const int lb = cls == 0 ? it.lm_begin[0] : cls == 1 ? it.lm_begin[1] : cls == 2 ? it.lm_begin[2]
               : cls == 3 ? it.lm_begin[3] : it.lm_begin[4];
  const int le = cls == 0 ? it.lm_end[0] : cls == 1 ? it.lm_end[1] : cls == 2 ? it.lm_end[2]
               : cls == 3 ? it.lm_end[3] : it.lm_end[4];
  const int tb = cls == 0 ? it.tile_begin[0] : cls == 1 ? it.tile_begin[1] : cls == 2 ? it.tile_begin[2]
               : cls == 3 ? it.tile_begin[3] : it.tile_begin[4];
  const size_t s = size_t(min(lb + ((T - tb) << (6 - sh)) + seg, le - 1));  // clamped: padding segments read a valid landmark
  const int64_t rw = cam >= 0 ? row : 0;
  const int cc = cam >= 0 ? cam : 0;
  jp_row<S>(p.JpS, p.JpT, rw, d.jp);
  d.vv = reinterpret_cast<const V4*>(p.Vh)[rw];
  const S* __restrict__ xc = x + 9 * cc;
#pragma unroll
  for (int c = 0; c < 9; ++c) d.xc[c] = xc[c];
  const int q = lane & 3;
  d.g = reinterpret_cast<const V4*>(p.givens + 16 * s)[q];
  d.rd0 = p.Rd[6 * s + q];
  d.rd1 = p.Rd[6 * s + min(4 + q, 5)];
  d.q1 = p.q1trd[3 * s + min(q, 2)];
  d.tau = p.tauH[3 * s + min(q, 2)];
  d.r00 = p.R0[6 * s + q];
  d.r01 = p.R0[6 * s + min(4 + q, 5)];
}

// quad-broadcast of lane Q's value (quad_perm:[Q,Q,Q,Q])
template <int Q, class S>
__device__ __forceinline__ S quad_bcast(S v) {
  return dpp_mov0<Q | (Q << 2) | (Q << 4) | (Q << 6)>(v);
}

template <class S, int P2>
__device__ __forceinline__ void bs_tile_compute(const Params<S>& p, const BsTileData<S>& d, int t_in_class,
                                                int lm_begin, int lm_end, int cam, int lane) {
  constexpr int LPW = 64 / P2;
  const int seg = lane / P2, r = lane - P2 * seg, base = lane - r;
  const int s = lm_begin + t_in_class * LPW + seg;
  const bool lm_ok = s < lm_end;
  const bool act = cam >= 0;
  // the landmark's scalars, whole, in every lane
  const S gc[6] = {quad_bcast<0>(d.g.x), quad_bcast<0>(d.g.y), quad_bcast<0>(d.g.z), quad_bcast<0>(d.g.w),
                   quad_bcast<1>(d.g.x), quad_bcast<1>(d.g.y)};
  const S gs[6] = {quad_bcast<1>(d.g.z), quad_bcast<1>(d.g.w), quad_bcast<2>(d.g.x), quad_bcast<2>(d.g.y),
                   quad_bcast<2>(d.g.z), quad_bcast<2>(d.g.w)};
  const S Rd[6] = {quad_bcast<0>(d.rd0), quad_bcast<1>(d.rd0), quad_bcast<2>(d.rd0), quad_bcast<3>(d.rd0),
                   quad_bcast<0>(d.rd1), quad_bcast<1>(d.rd1)};
  const S R0[6] = {quad_bcast<0>(d.r00), quad_bcast<1>(d.r00), quad_bcast<2>(d.r00), quad_bcast<3>(d.r00),
                   quad_bcast<0>(d.r01), quad_bcast<1>(d.r01)};
  const S q1[3] = {quad_bcast<0>(d.q1), quad_bcast<1>(d.q1), quad_bcast<2>(d.q1)};
  const S t0 = quad_bcast<0>(d.tau), t1 = quad_bcast<1>(d.tau), t2 = quad_bcast<2>(d.tau);
  S u = S(0);
#pragma unroll
  for (int c = 0; c < 9; ++c) u += d.jp[c] * d.xc[c];
  u = act ? u : S(0);
  const S v0 = act ? d.vv.x : S(0), v1 = act ? d.vv.y : S(0), v2 = act ? d.vv.z : S(0), qr = act ? d.vv.w : S(0);
  S t = u;
  t -= t0 * seg_sum<S, P2>(v0 * t) * v0;
  t -= t1 * seg_sum<S, P2>(v1 * t) * v1;
  t -= t2 * seg_sum<S, P2>(v2 * t) * v2;
  S tt[3] = {__shfl(t, base), __shfl(t, base + 1), __shfl(t, base + 2)};
  // landmark damping: the six rotations of stage 2 on (top rows, zero damping rows)
  {
    S dd[3] = {S(0), S(0), S(0)};
    int idx = 0;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
#pragma unroll
      for (int m = 0; m <= n; ++m) {
        const S cc = gc[idx], sn = gs[idx];
        const S xx = dd[n - m], yy = tt[n];
        dd[n - m] = cc * xx + sn * yy;
        tt[n] = -sn * xx + cc * yy;
        ++idx;
      }
    }
  }
  const S rhs0 = q1[0] + tt[0], rhs1 = q1[1] + tt[1], rhs2 = q1[2] + tt[2];
  S inc[3];
  inc[2] = rhs2 / Rd[5];
  inc[1] = (rhs1 - Rd[4] * inc[2]) / Rd[3];
  inc[0] = (rhs0 - Rd[1] * inc[1] - Rd[2] * inc[2]) / Rd[0];
  inc[0] = -inc[0];
  inc[1] = -inc[1];
  inc[2] = -inc[2];
  // w = Q^T (u + Jl delta) = t + [R0 delta; 0] (undamped triangle), model cost = sum w (w / 2 + Q^T r)
  S w = t;
  if (r == 0) w += R0[0] * inc[0] + R0[1] * inc[1] + R0[2] * inc[2];
  if (r == 1) w += R0[3] * inc[1] + R0[4] * inc[2];
  if (r == 2) w += R0[5] * inc[2];
  const S acc = seg_sum<S, P2>(act ? w * (S(0.5) * w + qr) : S(0));
  if (r == 0 && lm_ok) finish_landmark(p, s, inc, acc);
}

template <class S>
__global__ __launch_bounds__(256) void k_bs_tile(Params<S> p, ImplicitTiles it, const S* __restrict__ x) {
  stage_stamp(p.stamp);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int T = blockIdx.x * 4 + wave;
  if (T >= it.tile_begin[5]) return;
  int cam, row;
  tile_map(p, size_t(T), lane, cam, row);
  BsTileData<S> d;
  bs_tile_load(p, it, T, cam, row, lane, x, d);
  switch (hx_tile_class(it, T)) {
    case 0: bs_tile_compute<S, 4>(p, d, T - it.tile_begin[0], it.lm_begin[0], it.lm_end[0], cam, lane); break;
    case 1: bs_tile_compute<S, 8>(p, d, T - it.tile_begin[1], it.lm_begin[1], it.lm_end[1], cam, lane); break;
    case 2: bs_tile_compute<S, 16>(p, d, T - it.tile_begin[2], it.lm_begin[2], it.lm_end[2], cam, lane); break;
    case 3: bs_tile_compute<S, 32>(p, d, T - it.tile_begin[3], it.lm_begin[3], it.lm_end[3], cam, lane); break;
    default: bs_tile_compute<S, 64>(p, d, T - it.tile_begin[4], it.lm_begin[4], it.lm_end[4], cam, lane); break;
  }
}

// deterministic sum of the per-landmark model cost changes
__global__ __launch_bounds__(256) void k_sum_ldiff(const double* __restrict__ v, int n,
                                                   double* __restrict__ partials) {
  double acc = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) acc += v[i];
  __shared__ double sm[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double t = wave_sum(acc);
  if (lane == 0) sm[wave] = t;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// ===========================================================================
// Camera update: inc *= pose_scaling; T <- (exp(w) R, exp(w) t + v);
// intrinsics += inc[6..8]   (linearizor_qr.cpp:280-287, bal_problem.hpp:97-109)
// ===========================================================================
// cam = (q xyzw, t, f, k1, k2); inc = (dt, dphi, dintrinsics), already unscaled
template <class S>
__device__ __forceinline__ void retract_camera(S* cam, const S inc[9]) {
  // SO3::exp as a unit quaternion
  const S th2 = inc[3] * inc[3] + inc[4] * inc[4] + inc[5] * inc[5];
  S im, re;
  if (th2 < Eps<S>::eps * Eps<S>::eps) {
    const S th4 = th2 * th2;
    im = S(0.5) - S(1.0 / 48.0) * th2 + S(1.0 / 3840.0) * th4;
    re = S(1) - S(1.0 / 8.0) * th2 + S(1.0 / 384.0) * th4;
  } else {
    const S th = sqrt(th2);
    im = sin(S(0.5) * th) / th;
    re = cos(S(0.5) * th);
  }
  const S ax = im * inc[3], ay = im * inc[4], az = im * inc[5], aw = re;
  S dR[9];
  quat_to_rot(ax, ay, az, aw, dR);
  const S bx = cam[0], by = cam[1], bz = cam[2], bw = cam[3];
  S qx = aw * bx + ax * bw + ay * bz - az * by;
  S qy = aw * by - ax * bz + ay * bw + az * bx;
  S qz = aw * bz + ax * by - ay * bx + az * bw;
  S qw = aw * bw - ax * bx - ay * by - az * bz;
  const S nrm = S(1) / sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
  const S tx = cam[4], ty = cam[5], tz = cam[6];
  cam[0] = qx * nrm;
  cam[1] = qy * nrm;
  cam[2] = qz * nrm;
  cam[3] = qw * nrm;
  cam[4] = dR[0] * tx + dR[1] * ty + dR[2] * tz + inc[0];
  cam[5] = dR[3] * tx + dR[4] * ty + dR[5] * tz + inc[1];
  cam[6] = dR[6] * tx + dR[7] * ty + dR[8] * tz + inc[2];
  cam[7] += inc[6];
  cam[8] += inc[7];
  cam[9] += inc[8];
}

// `cams_bak` (rba_lm_step): the camera this update replaces is left there - BalProblem::backup() of the cameras
// (bal_problem.cpp:590-594) without a copy kernel of its own
template <class S>
__global__ void k_update_cameras(Params<S> p, const S* __restrict__ inc_scaled, S* __restrict__ cams_bak) {
  stage_stamp(p.stamp);
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p.n_cams) return;
  S inc[9], cam[10];
#pragma unroll
  for (int i = 0; i < 9; ++i) inc[i] = inc_scaled[9 * c + i] * p.pose_scaling[9 * c + i];
#pragma unroll
  for (int i = 0; i < 10; ++i) cam[i] = p.cams[10 * c + i];
  if (cams_bak) {
#pragma unroll
    for (int i = 0; i < 10; ++i) cams_bak[10 * c + i] = cam[i];
  }
  retract_camera<S>(cam, inc);
#pragma unroll
  for (int i = 0; i < 10; ++i) p.cams[10 * c + i] = cam[i];
}

// ---------------------------------------------------------------------------
// Mixed precision (RBA_MIXED): the optimisation state lives in double ("master" cameras / landmarks),
// the linear algebra of an LM iteration runs on its float rounding. The float increments are applied to
// the masters in double and the float state is re-rounded from them, so the state never accumulates
// float rounding from one iteration to the next, and costs are evaluated in double on the masters.
// ---------------------------------------------------------------------------
__global__ void k_mixed_update_cameras(double* __restrict__ cams64, float* __restrict__ cams32,
                                       const float* __restrict__ inc_scaled, const float* __restrict__ pose_scaling,
                                       int n_cams, unsigned long long* __restrict__ stamp) {
  stage_stamp(stamp);
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cams) return;
  double inc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) inc[i] = double(inc_scaled[9 * c + i]) * double(pose_scaling[9 * c + i]);
  double* cam = cams64 + 10 * c;
  retract_camera<double>(cam, inc);
#pragma unroll
  for (int i = 0; i < 10; ++i) cams32[10 * c + i] = float(cam[i]);
}

__global__ void k_mixed_update_landmarks(double* __restrict__ lms64, float* __restrict__ lms32,
                                         const float* __restrict__ lm_inc, int64_t n) {
  const int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const double v = lms64[i] + double(lm_inc[i]);
  lms64[i] = v;
  lms32[i] = float(v);
}

__global__ void k_mixed_round(const double* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i < n) dst[i] = float(src[i]);
}

// ===========================================================================
// Block-diagonal preconditioner (BlockDiagonalPreconditioner,
// src/rootba/cg/preconditioner.hpp:79-136): one thread per camera inverts its
// SPD 9x9 block by Cholesky (upper triangle is the definition, App. A.5).
// A block with a pivot that is not positive - a float32 block whose smallest eigenvalue sits below its rounding error,
// met on final-13682 at lambda ~ 1e-6 - is replaced by its DIAGONAL (a point-Jacobi preconditioner for that camera):
// Eigen's LLT stops at such a pivot and leaves a meaningless but finite factor (the reference then preconditions that
// camera with garbage), while the square root of a negative pivot would put NaNs into every PCG vector and end the
// solve at its first iteration with a zero camera increment (observed: seven LM iterations in a row moved landmarks
// only). Any SPD matrix is a valid preconditioner; blocks that factor are untouched; bit 4 of the failure word records
// that it happened. (Raising the pivot to the rounding level of its diagonal entry instead was tried: the huge inverse
// entries along the near-null direction produce wild steps.)
// ===========================================================================
// SB = scalar of the blocks and of the factorisation (double blocks of a float solver: kernels_a64.hpp, k_a64_diag<true>),
// `damp` is added to their diagonal; the inverse is stored in the solver scalar.
template <class S, class SB = S>
__global__ __launch_bounds__(64) void k_invert_blocks(const SB* __restrict__ blocks, S* __restrict__ inv, int n_cams,
                                int* fail_flag, SB damp = SB(0), unsigned long long* __restrict__ stamp = nullptr) {
  stage_stamp(stamp);
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cams) return;
  using T = SB;
  T L[45];  // lower triangle, row-major packed: L(i,j) at i(i+1)/2 + j
  const SB* __restrict__ src = blocks + 81 * c;
  auto a = [&](int e) { return T(src[e]) + (e % 10 == 0 ? damp : T(0)); };
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    T d = a(9 * j + j);
#pragma unroll
    for (int q = 0; q < j; ++q) d -= L[j * (j + 1) / 2 + q] * L[j * (j + 1) / 2 + q];
    ok = ok && (d > T(0));
    const T ljj = sqrt(d);
    L[j * (j + 1) / 2 + j] = ljj;
    const T ij = T(1) / ljj;
#pragma unroll
    for (int i = j + 1; i < 9; ++i) {
      T v = a(9 * j + i);  // upper triangle entry (j,i)
#pragma unroll
      for (int q = 0; q < j; ++q) v -= L[i * (i + 1) / 2 + q] * L[j * (j + 1) / 2 + q];
      L[i * (i + 1) / 2 + j] = v * ij;
    }
  }
  S* out = inv + 81 * c;
  if (!ok) {
    atomicOr(fail_flag, 4);
    bool diag_ok = true;
#pragma unroll
    for (int j = 0; j < 9; ++j) diag_ok = diag_ok && (a(9 * j + j) > T(0)) && is_finite(a(9 * j + j));
    if (diag_ok) {  // (else: NaN / non-positive diagonal - a numerical failure upstream, left to propagate)
#pragma unroll
      for (int e = 0; e < 81; ++e) out[e] = (e % 10 == 0) ? S(T(1) / a(e)) : S(0);
      return;
    }
  }
#pragma unroll
  for (int col = 0; col < 9; ++col) {
    T yv[9], xv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      T v = (i == col) ? T(1) : T(0);
#pragma unroll
      for (int q = 0; q < i; ++q) v -= L[i * (i + 1) / 2 + q] * yv[q];
      yv[i] = v / L[i * (i + 1) / 2 + i];
    }
#pragma unroll
    for (int i = 8; i >= 0; --i) {
      T v = yv[i];
#pragma unroll
      for (int q = i + 1; q < 9; ++q) v -= L[q * (q + 1) / 2 + i] * xv[q];
      xv[i] = v / L[i * (i + 1) / 2 + i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) out[9 * i + col] = S(xv[i]);
  }
}

// ===========================================================================
// PCG (ConjugateGradientsSolver::solve, src/rootba/cg/conjugate_gradient.hpp:113-298)
// Five small multi-workgroup kernels per iteration around H*x:
//   k_pcg_a1 : z = M^-1 r, per-block partial of rho = r.z
//   k_pcg_a2 : rho, beta (every block sums the partials in the same fixed order),
//              p = z + beta p, q = 0
//   k_pcg_b1 : q += lambda p, partial of pq = p.q
//   k_pcg_b2 : alpha, x += alpha p, r -= alpha q, partial of Q = -x.(b + r)
//              (or, every 10th iteration, prepares the residual refresh)
//   k_pcg_fin: Q-model termination test, iteration counter
//   k_pcg_c1 : (refresh) r = b - H x, partial of Q
// One workgroup cannot pull the 81 n_c preconditioner through a single CU fast
// enough (H*x evicts it from L2 every iteration), hence kPcgBlocks workgroups.
// (Measured alternative: the five kernels fused into one launch with software grid
//  barriers between the phases. Same 0.65 ms per CG iteration on venice — with one L2
//  per XCD an agent-scope release/acquire pair is an L2 write-back + invalidate, i.e.
//  as expensive as the ~5.6 us kernel boundary it replaces — so the simpler form stays.
//  Likewise measured and dropped: the termination test in the last-arriving workgroup of
//  k_pcg_b2 (threadfence + atomic counter) and k_pcg_b1 folded into the SpMV: slower, for
//  the same reason; operand loads hoisted above the `done` test: +1 %, not worth the code;
//  all vector work of an iteration in ONE 1024-thread workgroup (no grid synchronisation at
//  all): 50 us instead of 5 x 4.8 us — a single CU cannot stream 0.9 MB of vectors and
//  preconditioner blocks fast enough; only b1 + b2 + the termination test in one workgroup:
//  still 10 us per iteration slower; three regrouped multi-workgroup launches per iteration
//  (direction + ping-ponged scalar state | S p with p.q | update + preconditioner per owned
//  cameras): 48 us instead of 44 us. The iteration is a chain of ~25 dependent memory round
//  trips through caches that every launch starts cold, not a count of launches.)
// All scalars stay on the device in `CgState` (double, as in the reference);
// kernels are no-ops once `done`; the host only polls the state. Every reduction
// has a fixed order, so all ranks of a multi-GPU run compute bit-identical
// scalars from the (identical) all-reduced vectors.
// ===========================================================================
struct CgState {
  double rho_hist[2];  // rho of iteration i (0-based) lives in rho_hist[i & 1]
  double q_hist[2];    // Q after i completed iterations lives in q_hist[i & 1] (Q_0 = 0)
  double pq, norm_b2;
  double alpha, beta;
  int iter;         // iterations completed (written by k_pcg_fin / k_pcgs_update only)
  int cur;          // iteration in progress, 1-based (written by k_pcgs_spmv only)
  int need_test;    // k_pcgs_update left Q partials of iteration `iter` for the termination test
  int done;         // 0 running, 1 finished
  int termination;  // 0 NO_CONVERGENCE, 1 SUCCESS, 2 FAILURE
  int result_iter;  // num_iterations of the summary (valid once done)
  int indefinite;   // terminated on p.q <= 0 ("Matrix is indefinite")
  int refresh;      // the current iteration recomputes r from scratch
  // per-solve parameters of the fused PCG (kernels_pcg.hpp), set by k_pcgs_begin: device-side so
  // that the captured launch graphs are independent of them
  int pswap;        // direction of `it` completed iterations lives in buffer (it + pswap) & 1
  double lambda;    // pose damping added to the product (0 when the matrix contains it)
};

constexpr int kPcgBlocks = 64;  // workgroups of the PCG vector kernels
constexpr int kPcgThreads = 256;

// deterministic sum over the workgroup (4 waves); result valid in all threads
__device__ __forceinline__ double pcg_block_sum(double v, double* sm) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double t = wave_sum(v);
  __syncthreads();
  if (lane == 0) sm[wave] = t;
  __syncthreads();
  return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// Verification of a solve that ran on the ASSEMBLED float32 matrix (Solver::solve): with hx = (sum_l A_l^T A_l) x from
// the matrix-free operator, the Q model -x.(b + r) / 2 is evaluated once with the recursion's residual r and once with
// the true one b - (hx + lambda x). out = { x.(b + r_rec), x.(b + r_true), |r_true|^2, |r_rec|^2 }   (single workgroup)
template <class S>
__global__ __launch_bounds__(1024) void k_solve_check(const S* __restrict__ x, const S* __restrict__ bvec,
                                                     const S* __restrict__ r, const S* __restrict__ hx, S lambda, int n,
                                                     double* __restrict__ out) {
  __shared__ double sm[16][4];
  double acc[4] = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double xi = double(x[i]), bi = double(bvec[i]), ri = double(r[i]);
    const double rt = bi - (double(hx[i]) + double(lambda) * xi);
    acc[0] += xi * (bi + ri);
    acc[1] += xi * (bi + rt);
    acc[2] += rt * rt;
    acc[3] += ri * ri;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double t = wave_sum(acc[k]);
    if (lane == 0) sm[wave][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double v = 0;
    for (int w = 0; w < int(blockDim.x >> 6); ++w) v += sm[w][threadIdx.x];
    out[threadIdx.x] = v;
  }
}

// x = 0, r = b, state reset, |b|^2   (single workgroup)
template <class S>
__global__ __launch_bounds__(1024) void k_pcg_init(const S* __restrict__ bvec, S* __restrict__ x,
                                                  S* __restrict__ r, int n, CgState* st,
                                                  unsigned long long* __restrict__ stamp) {
  stage_stamp(stamp);
  __shared__ double sm[16];
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const S v = bvec[i];
    x[i] = S(0);
    r[i] = v;
    acc += double(v) * double(v);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double t = wave_sum(acc);
  if (lane == 0) sm[wave] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    double nb2 = 0;
    for (int w = 0; w < int(blockDim.x >> 6); ++w) nb2 += sm[w];
    st->rho_hist[0] = st->rho_hist[1] = 1.0;
    st->pq = 0;
    st->q_hist[0] = st->q_hist[1] = 0;  // -x.(b + r) with x = 0
    st->norm_b2 = nb2;
    st->alpha = st->beta = 0;
    st->iter = 0;
    st->cur = 0;
    st->need_test = 0;
    st->result_iter = 0;
    st->indefinite = 0;
    st->refresh = 0;
    st->termination = nb2 == 0.0 ? 1 : 0;  // "Convergence. |b| = 0."
    st->done = nb2 == 0.0 ? 1 : 0;
  }
}

template <class S>
__global__ __launch_bounds__(kPcgThreads) void k_pcg_a1(const S* __restrict__ inv,
                                                       const S* __restrict__ r, S* __restrict__ z,
                                                       int n, const CgState* st,
                                                       double* __restrict__ partial) {
  __shared__ double sm[4];
  if (st->done) return;
  double acc = 0;
  for (int i = blockIdx.x * kPcgThreads + threadIdx.x; i < n; i += kPcgBlocks * kPcgThreads) {
    const int c = i / 9, row = i - 9 * c;
    const S* M = inv + 81 * c + 9 * row;
    const S* rc = r + 9 * c;
    S v = S(0);
#pragma unroll
    for (int j = 0; j < 9; ++j) v += M[j] * rc[j];
    z[i] = v;
    acc += double(r[i]) * double(v);
  }
  const double t = pcg_block_sum(acc, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// (The seven-kernel PCG loop of round 1 - k_block_apply, k_series_step, k_pcg_rho / a2 / b1 / b2 / c1 / fin - served the
//  matrix-free power-series solves until round 6; they run in the two-launch protocol of kernels_pcg.hpp now.)

// PMC calibration: stream `n` floats once with dword (VEC = 1) or 16-byte
// (VEC = 4) loads — a known byte count in this code's own access patterns, to
// calibrate rocprofv3's FETCH_SIZE on gfx950 (MI355X_MICROARCH.md §HBM).
template <int VEC>
__global__ __launch_bounds__(256) void k_calib_read(const float* __restrict__ src, size_t n,
                                                    float* __restrict__ sink) {
  float acc = 0.f;
  const size_t stride = size_t(gridDim.x) * 256 * VEC;
  for (size_t i = (size_t(blockIdx.x) * 256 + threadIdx.x) * VEC; i + VEC <= n; i += stride) {
    if (VEC == 4) {
      const float4 v = *reinterpret_cast<const float4*>(src + i);
      acc += v.x + v.y + v.z + v.w;
    } else {
      acc += src[i];
    }
  }
  if (acc == 123.456f) sink[0] = acc;  // keep the loads alive
}

// blocks[c](d,d) -= excess  (multi-GPU: lambda*I was added once per rank)
template <class S>
__global__ void k_sub_diag(S* __restrict__ blocks, S excess, int n_cams) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 9 * n_cams) blocks[81 * (i / 9) + 10 * (i % 9)] -= excess;
}

// out[i] = d[i] * v[i] (out may alias v): the pose scaling applied to a camera-sized vector (compact stage 2)
template <class S>
__global__ void k_scale_vec(const S* __restrict__ v, const S* __restrict__ d, S* __restrict__ out, int n,
                            unsigned long long* __restrict__ stamp) {
  stage_stamp(stamp);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = d[i] * v[i];
}

template <class S>
__global__ void k_negate(S* __restrict__ v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = -v[i];
}

// End of a solve inside the LM loop (rba_lm_step): x = -x stays on the device - a copy goes to the buffer the
// back-substitution reads - and the host gets what it needs to judge the step, |x|^2 and the number of non-finite
// entries, in its pinned page instead of the vector itself (single workgroup).
// Since round 6 the same workgroup does what two more launches of ~5 us did:
//   `xs`        D inc, the operand of the back-substitution on the unscaled Jacobian rows (k_scale_vec);
//   `flag_host` the failure word of the linearisation this solve belongs to (k_publish_flag), bits `flag_clear` reset.
// (Measured and dropped: the camera update of the step in here too - 1778 retractions by one workgroup took the kernel
//  from 8 to 22 us where k_update_cameras on 28 workgroups takes 5; the reductions that end the cost evaluation and the
//  back-substitution in the LAST workgroup of their producers (ticket + __threadfence): 32 -> 260 us and 5 -> 130 us -
//  an agent-scope fence per workgroup is an L2 write-back on this part; the landmark backup inside the
//  back-substitution: + 7 us there for a 6 us copy. gpurun_out/r6b, profiles/r6b_fusion_that_did_not_pay_kernel_stats.csv)
template <class S>
__global__ __launch_bounds__(1024) void k_finish_increment(S* __restrict__ x, S* __restrict__ inc, int n,
                                                          double* __restrict__ out_host, const CgState* st,
                                                          CgState* st_host, const S* __restrict__ pose_scaling,
                                                          S* __restrict__ xs, int* __restrict__ flag,
                                                          int* __restrict__ flag_host, int flag_clear) {
  __shared__ double sm[16][2];
  if (st_host) {  // the final PCG state, if the host has not read it yet
    constexpr int kWords = int(sizeof(CgState) / sizeof(int));
    static_assert(kWords <= 64 && sizeof(CgState) % sizeof(int) == 0, "copied as words by one wave");
    if (threadIdx.x < kWords) reinterpret_cast<int*>(st_host)[threadIdx.x] = reinterpret_cast<const int*>(st)[threadIdx.x];
  }
  if (flag_host && threadIdx.x == 1023) {
    const int f = *flag;
    *flag_host = f;
    if (f & flag_clear) *flag = f & ~flag_clear;
  }
  double acc = 0, bad = 0;
  // (batches of eight entries per work-item, every load of a batch requested before its first store: as a plain strided
  //  loop the sixteen entries of a work-item were sixteen memory round trips in a row - x is read and written)
  for (int base = 0; base < n; base += 8 * 1024) {
    S xv[8], ps[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int ic = min(base + u * 1024 + int(threadIdx.x), n - 1);
      xv[u] = x[ic];
      ps[u] = xs ? pose_scaling[ic] : S(0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 1024 + int(threadIdx.x);
      if (i < n) {
        const S v = -xv[u];
        x[i] = v;
        inc[i] = v;
        if (xs) xs[i] = ps[u] * v;
        acc += double(v) * double(v);
        bad += is_finite(v) ? 0.0 : 1.0;
      }
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double t0 = wave_sum(acc), t1 = wave_sum(bad);
  if (lane == 0) {
    sm[wave][0] = t0;
    sm[wave][1] = t1;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    double v = 0;
    for (int w = 0; w < 16; ++w) v += sm[w][threadIdx.x];
    out_host[threadIdx.x] = v;
  }
}

template <class S>
__global__ void k_axpy_lambda(const S* __restrict__ x, S* __restrict__ y, S lambda, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += lambda * x[i];
}

}  // namespace rba
