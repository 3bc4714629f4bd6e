// kernels_s1.hpp — the landmark side of stage 1 (linearise + marginalise, LinearizationQR::get_stage1,
// src/rootba/qr/linearization_qr.hpp:634-712) and of stage 2 (get_stage2, :716-815), staged BY PARALLELISM.
// The dense 2k x 9k block of a landmark is never materialised.
//
// Round 1 ran geometry, Householder QR and the column pass inside ONE wavefront-per-landmark(s)
// kernel: 86 % VALU-busy (profiles/r2_pmc_stage1_round1_kernels.csv) at 5-40 % lane utilisation, 1.5 ms on
// venice-1778. Here every pass runs at the parallelism its work has:
//
//   k_s1_geometry   one THREAD per observation: projection, analytic Jacobians, Huber weight
//                   (linearize_landmark, landmark_block_base.ipp:88-147). Writes the weighted pose
//                   Jacobian rows (JpS, unscaled) and the rows [sqrt(w) Jl | sqrt(w) r] (Vh),
//                   transposed through LDS so that both stores are contiguous 16-byte streams.
//   (camera-major Gram pass: kernels_cam.hpp)
//   k_s1_qr_tile    one LANE per block row, a landmark = an aligned group of 4..64 lanes (the wave
//                   tiles of the implicit-Q operator): Jl column scaling (scale_Jl_cols, ipp:571-587),
//                   Householder QR of the 2k x 3 Jl (perform_qr_householder, ipp:717-743), Q^T r, and the
//                   per-landmark scalars of the compact reflector application. Only the 4 columns
//                   [Jl | r] are transformed here - Q depends on Jl alone.
//   k_s1_fused_obs  (round 4, the default for the wave-tile landmarks) geometry AND QR in one kernel with one lane per
//                   OBSERVATION: [Jl | r] stays in registers between the two, a landmark is an aligned group of
//                   2..32 lanes; k_s1_geometry then serves only the longer tracks, k_s1_qr_tile the sub-stage timers.
//   k_s2_obs        stage 2, one THREAD per observation: the six damping rotations of its landmark
//                   (set_landmark_damping, ipp:165-210) and the observation's stage-2 record WA.
//   k_s12_cols      one THREAD per observation, ON DEMAND (assembly of the reduced matrix, matrix-free E0
//                   products): column scaling (scale_Jp_cols, ipp:589-614, commutes with Q^T), the three damped
//                   top rows Q1^T Jp and the Q2 part of b in CLOSED FORM: with
//                   Q^T Jp[:, j] = Jp[:, j] - sum_m c_m v_m (three reflectors, c_m from two FMAs each because
//                   column j has two non-zero rows),
//                     b_j = sum_{r >= 3} (Q^T Jp)[r, j] (Q^T r)[r]
//                         = m0 q[2i] + m1 q[2i+1] (rows >= 3 only) - sum_m c_m d_m,   d_m = sum_{r>=3} v_m[r] q[r]
//                   - the same products as the row-by-row sum, associated per reflector, O(1) per
//                   column instead of O(2k).
// Landmarks with more than 112 observations keep a workgroup-per-landmark QR (kernels_big.hpp).
#pragma once

#include "kernels.hpp"
#include "kernels_cam.hpp"

namespace rba {

// ---------------------------------------------------------------------------
// pass G
// ---------------------------------------------------------------------------
template <class S>
__global__ __launch_bounds__(256) void k_s1_geometry(Params<S> p, int64_t o_begin, int64_t n_obs) {
  stage_stamp(p.stamp);
  extern __shared__ __attribute__((aligned(16))) char smem_s1[];
  S* sj = reinterpret_cast<S*>(smem_s1);  // [256][18]
  S* sv = sj + 256 * 18;                  // [256][8]
  const int tid = threadIdx.x;
  const int64_t o_base = o_begin + int64_t(blockIdx.x) * 256;  // (o_begin: a multiple of 2 - 16-byte aligned streams)
  const int64_t o = o_base + tid;
  const int n_here = int(min<int64_t>(256, n_obs - o_base));
  if (tid < n_here) {
    const int cam = p.obs_cam[o];
    const int l = p.obs_lm[o];
    S res[2], Jp[18], Jl[6];
    const bool valid = linearize_obs<S>(p.cams + 10 * cam, p.lms[3 * l], p.lms[3 * l + 1], p.lms[3 * l + 2],
                                        p.obs_xy[2 * o], p.obs_xy[2 * o + 1], res, Jp, Jl);
    S sw = S(0);
    if (!p.valid_only || valid) {
      bool fin = is_finite(res[0]) && is_finite(res[1]);
#pragma unroll
      for (int i = 0; i < 18; ++i) fin = fin && is_finite(Jp[i]);
#pragma unroll
      for (int i = 0; i < 6; ++i) fin = fin && is_finite(Jl[i]);
      if (!fin) atomicOr(p.fail_flag, 1);  // non-finite check of linearize_landmark (ipp:123-146)
      S err, w;
      error_weight<S>(p.robust_norm, p.huber, res[0] * res[0] + res[1] * res[1], err, w);
      sw = sqrt(w);
    }
#pragma unroll
    for (int c = 0; c < 18; ++c) Jp[c] *= sw;
#pragma unroll
    for (int c = 0; c < 18; ++c) sj[18 * tid + c] = Jp[c];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      sv[8 * tid + 4 * r + 0] = sw * Jl[3 * r + 0];
      sv[8 * tid + 4 * r + 1] = sw * Jl[3 * r + 1];
      sv[8 * tid + 4 * r + 2] = sw * Jl[3 * r + 2];
      sv[8 * tid + 4 * r + 3] = sw * res[r];
    }
  }
  __syncthreads();
  // contiguous copy-out (observations of a workgroup are consecutive)
  using V = typename std::conditional<sizeof(S) == 4, float4, double2>::type;
  constexpr int N = 16 / int(sizeof(S));
  jp_store_rows<S>(p.JpS, p.JpT, o_base, n_here, sj, tid, 256);
  {
    S* dst = p.Vh + 8 * o_base;
    const int nvec = 8 * n_here / N;
    for (int i = tid; i < nvec; i += 256) reinterpret_cast<V*>(dst)[i] = reinterpret_cast<const V*>(sv)[i];
  }
}

// B_mid = D G D once the (all-reduced) column scaling is known
template <class S>
__global__ void k_scale_gram(Params<S> p) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 81 * p.n_cams) return;
  const int c = t / 81, e = t - 81 * c, a = e / 9, b = e - 9 * a;
  p.B_mid[t] *= p.pose_scaling[9 * c + a] * p.pose_scaling[9 * c + b];
}

// ---------------------------------------------------------------------------
// pass Q: Householder QR of [Jl | r], lane per block row
// ---------------------------------------------------------------------------
// per-landmark scalars handed to the column pass: LQ[s][12] = tau[3], g10 g20 g21, d[3], pad
template <class S, int P2>
__device__ __forceinline__ void s1_qr_tile(const Params<S>& p, size_t T, int t_in_class, int lm_begin, int lm_end,
                                           int lane) {
  using V4 = typename std::conditional<sizeof(S) == 4, float4, double4>::type;
  constexpr int LPW = 64 / P2;
  const int seg = lane / P2, r = lane - P2 * seg, base = lane - r;
  const int s = lm_begin + t_in_class * LPW + seg;
  const bool lm_ok = s < lm_end;
  int cam_unused, row_i;
  tile_map(p, T, lane, cam_unused, row_i);
  const int64_t row = row_i;  // -1: padding lane
  const bool rvalid = row >= 0;
  S jl[3] = {S(0), S(0), S(0)}, rs = S(0);
  if (rvalid) {
    const V4 v = reinterpret_cast<const V4*>(p.Vh)[row];
    jl[0] = v.x;
    jl[1] = v.y;
    jl[2] = v.z;
    rs = v.w;
  }
  // scale_Jl_cols
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const S ss = seg_sum<S, P2>(jl[c] * jl[c]);
    const S sc = fast_rcp(p.eps + fast_sqrt(ss));
    jl[c] *= sc;
    if (r == 0 && lm_ok) p.jl_scale[3 * s + c] = sc;
  }
  // (the pre-QR rows are not kept for the wave tiles: their back-substitution evaluates the model cost in the
  //  rotated frame, k_bs_tile; the wide / big kernels below keep JlS / rS for the two-kernel back-substitution)
  S vm[3], tau[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const S c0 = __shfl(jl[m], base + m);
    const S tail = seg_sum<S, P2>((r > m && rvalid) ? jl[m] * jl[m] : S(0));
    S beta, inv;
    if (tail <= Eps<S>::tiny) {
      tau[m] = S(0);
      beta = c0;
      inv = S(0);
    } else {
      // (IEEE operations here: beta, the reflector's scaling and tau decide how orthogonal H = I - tau v v^T is, and the
      //  1-ulp hardware reciprocal / square root doubled the distance of a 137-iteration matrix-free solve on an
      //  ill-conditioned final-13682 state from the float64 iterate - 3.1e-2 against the 1.2e-2 of the float32 CPU run)
      beta = sqrt(c0 * c0 + tail);
      if (c0 >= S(0)) beta = -beta;
      inv = S(1) / (c0 - beta);
      tau[m] = (beta - c0) / beta;
    }
    vm[m] = (r == m) ? S(1) : ((r > m && rvalid) ? jl[m] * inv : S(0));
#pragma unroll
    for (int c2 = m + 1; c2 < 3; ++c2) {
      const S d = tau[m] * seg_sum<S, P2>(vm[m] * jl[c2]);
      jl[c2] -= d * vm[m];
    }
    {
      const S d = tau[m] * seg_sum<S, P2>(vm[m] * rs);
      rs -= d * vm[m];
    }
    if (r == m) jl[m] = beta;
    if (r > m) jl[m] = S(0);
  }
  const S g10 = seg_sum<S, P2>(vm[1] * vm[0]), g20 = seg_sum<S, P2>(vm[2] * vm[0]),
          g21 = seg_sum<S, P2>(vm[2] * vm[1]);
  const bool low = r >= 3 && rvalid;
  const S d0 = seg_sum<S, P2>(low ? vm[0] * rs : S(0)), d1 = seg_sum<S, P2>(low ? vm[1] * rs : S(0)),
          d2 = seg_sum<S, P2>(low ? vm[2] * rs : S(0));
  const S r00 = __shfl(jl[0], base), r01 = __shfl(jl[1], base), r02 = __shfl(jl[2], base),
          r11 = __shfl(jl[1], base + 1), r12 = __shfl(jl[2], base + 1), r22 = __shfl(jl[2], base + 2);
  if (r == 0 && lm_ok) {
    S* R = p.R0 + 6 * s;
    R[0] = r00;
    R[1] = r01;
    R[2] = r02;
    R[3] = r11;
    R[4] = r12;
    R[5] = r22;
    p.tauH[3 * s + 0] = tau[0];
    p.tauH[3 * s + 1] = tau[1];
    p.tauH[3 * s + 2] = tau[2];
    V4* lq = reinterpret_cast<V4*>(p.LQ + 12 * size_t(s));
    lq[0] = V4{tau[0], tau[1], tau[2], g10};
    lq[1] = V4{g20, g21, d0, d1};
    lq[2] = V4{d2, S(0), S(0), S(0)};
  }
  if (rvalid) reinterpret_cast<V4*>(p.Vh)[row] = V4{vm[0], vm[1], vm[2], rs};
}

template <class S>
__global__ __launch_bounds__(256) void k_s1_qr_tile(Params<S> p, ImplicitTiles it) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int T = blockIdx.x * 4 + wave;
  if (T >= it.tile_begin[5]) return;
  if (T >= it.tile_begin[4])
    s1_qr_tile<S, 64>(p, T, T - it.tile_begin[4], it.lm_begin[4], it.lm_end[4], lane);
  else if (T >= it.tile_begin[3])
    s1_qr_tile<S, 32>(p, T, T - it.tile_begin[3], it.lm_begin[3], it.lm_end[3], lane);
  else if (T >= it.tile_begin[2])
    s1_qr_tile<S, 16>(p, T, T - it.tile_begin[2], it.lm_begin[2], it.lm_end[2], lane);
  else if (T >= it.tile_begin[1])
    s1_qr_tile<S, 8>(p, T, T - it.tile_begin[1], it.lm_begin[1], it.lm_end[1], lane);
  else
    s1_qr_tile<S, 4>(p, T, T - it.tile_begin[0], it.lm_begin[0], it.lm_end[0], lane);
}

// ---------------------------------------------------------------------------
// passes G + Q in one kernel, an OBSERVATION (two block rows) per lane
// ---------------------------------------------------------------------------
// The QR kernel above is bound by its instruction count, and fusing the geometry into it with a lane per block ROW
// (tried: 224 us against 262 for the two kernels on venice-1778) evaluates the projection in both lanes of an
// observation while every segment reduction serves 2k lanes. Here a lane holds both rows of its observation: the geometry runs once per observation, a landmark is an aligned group of P = 1/2 P2 lanes
// (one reduction step less), and a wavefront covers TWO consecutive row tiles of a class (lane j: row tile
// 2 t + (j >> 5), its lanes 2 (j & 31) and 2 (j & 31) + 1) - the lane maps of the row tiles (RT, CT) serve both.
// Same operations per value as the two-kernel stage 1 except that the two rows of a lane are added before the
// cross-lane part of a sum.
template <class S, int P>
__device__ __forceinline__ void s1_fused_obs(const Params<S>& p, int T0, int n_tiles_class, int t_pair, int lm_begin,
                                             int lm_end, int lane, S* __restrict__ stage) {
  using V4 = typename std::conditional<sizeof(S) == 4, float4, double4>::type;
  constexpr int P2 = 2 * P, LPW = 64 / P2;  // (row-tile quantities)
  const int h = lane >> 5, q = lane & 31;
  const int t_in_class = 2 * t_pair + h;
  const bool tile_ok = t_in_class < n_tiles_class;
  const size_t T = size_t(T0) + size_t(tile_ok ? t_in_class : 2 * t_pair);
  const int seg = q / P, r = q - P * seg, base = lane - r;
  const int s = lm_begin + t_in_class * LPW + seg;
  const bool lm_ok = tile_ok && s < lm_end;
  // {camera, first block row} of the lane's observation slot (unconditional loads of valid addresses, then masked; the
  // row - -1 in a padding slot - here, the camera where it is used: as ONE 8-byte load the pair stayed live across the
  // kernel's register peak, 66 registers instead of 64 = one wavefront per SIMD less, + 6 us)
  const int* __restrict__ slot = reinterpret_cast<const int*>(p.OT + (T * 32 + q));
  const int row_of_tile = slot[1];
  const int64_t row = tile_ok ? int64_t(row_of_tile) : int64_t(-1);  // first row of the observation; -1: padding
  const bool valid_lane = row >= 0;
  const uint64_t live = __ballot(valid_lane);
  const int first = __builtin_ctzll(live | (uint64_t(1) << 63));
  const int64_t o_first = __shfl(row, first) >> 1;
  S ja[3], jb[3], ra, rb;
  {
    // branch-free (padding lanes evaluate a clamped observation and are masked at the end): inside a conditional the
    // compiler sinks the index and point loads behind the wait for `row` - three round trips instead of two
    const int cam = max(slot[0], 0);
    const S* __restrict__ lp = p.lms + 3 * size_t(min(max(s, lm_begin), lm_end - 1));
    const int64_t o = (valid_lane ? row : int64_t(0)) >> 1;
    S res[2], Jp[18], Jl[6];
    const bool valid = linearize_obs<S>(p.cams + 10 * cam, lp[0], lp[1], lp[2], p.obs_xy[2 * o], p.obs_xy[2 * o + 1],
                                        res, Jp, Jl);
    S sw = S(0);
    if (!p.valid_only || valid) {
      // non-finite check of linearize_landmark (ipp:123-146): 0 * v is 0 for every finite v and NaN otherwise - one
      // multiply-add per value and one test instead of a class test per value
      S z = S(0) * res[0] + S(0) * res[1];
#pragma unroll
      for (int i = 0; i < 18; ++i) z += S(0) * Jp[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) z += S(0) * Jl[i];
      if (!(z == S(0)) && valid_lane) atomicOr(p.fail_flag, 1);
      S err, w;
      error_weight<S>(p.robust_norm, p.huber, res[0] * res[0] + res[1] * res[1], err, w);
      sw = sqrt(w);
    }
    if (!valid_lane) sw = S(0);
    if (valid_lane) {
      // (staged in the split storage of the rows - kernels.hpp, jp_row: eight entries per row, the ninth behind the 64
      //  observations' main parts - so that the copy-out below is two straight streams)
      const int oi = int(o - o_first);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int c = 0; c < 8; ++c) stage[16 * oi + 8 * r + c] = sw * Jp[9 * r + c];
        stage[16 * 64 + 2 * oi + r] = sw * Jp[9 * r + 8];
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ja[c] = valid_lane ? sw * Jl[c] : S(0);
      jb[c] = valid_lane ? sw * Jl[3 + c] : S(0);
    }
    ra = valid_lane ? sw * res[0] : S(0);
    rb = valid_lane ? sw * res[1] : S(0);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  {
    // (the wavefront's observations are consecutive: two coalesced streams)
    const int n = __popcll(live);
    V4* __restrict__ dst = reinterpret_cast<V4*>(p.JpS) + 4 * o_first;
    const V4* __restrict__ src = reinterpret_cast<const V4*>(stage);
    for (int i = lane; i < 4 * n; i += 64) dst[i] = src[i];
    for (int i = lane; i < 2 * n; i += 64) p.JpT[2 * o_first + i] = stage[16 * 64 + i];
  }
  // scale_Jl_cols
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const S ss = seg_sum<S, P>(ja[c] * ja[c] + jb[c] * jb[c]);
    const S sc = fast_rcp(p.eps + fast_sqrt(ss));
    ja[c] *= sc;
    jb[c] *= sc;
    if (r == 0 && lm_ok) p.jl_scale[3 * s + c] = sc;
  }
  // Householder QR of the 2k x 3 block, rows 2r (a) and 2r + 1 (b) in this lane
  const int row_a = 2 * r, row_b = 2 * r + 1;
  S va[3], vb[3], tau[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const S c0 = __shfl((m & 1) ? jb[m] : ja[m], base + (m >> 1));
    const S tail = seg_sum<S, P>(((row_a > m && valid_lane) ? ja[m] * ja[m] : S(0)) +
                                 ((row_b > m && valid_lane) ? jb[m] * jb[m] : S(0)));
    S beta, inv;
    if (tail <= Eps<S>::tiny) {
      tau[m] = S(0);
      beta = c0;
      inv = S(0);
    } else {
      // (IEEE operations: see s1_qr_tile)
      beta = sqrt(c0 * c0 + tail);
      if (c0 >= S(0)) beta = -beta;
      inv = S(1) / (c0 - beta);
      tau[m] = (beta - c0) / beta;
    }
    va[m] = (row_a == m) ? S(1) : ((row_a > m && valid_lane) ? ja[m] * inv : S(0));
    vb[m] = (row_b == m) ? S(1) : ((row_b > m && valid_lane) ? jb[m] * inv : S(0));
#pragma unroll
    for (int c2 = m + 1; c2 < 3; ++c2) {
      const S d = tau[m] * seg_sum<S, P>(va[m] * ja[c2] + vb[m] * jb[c2]);
      ja[c2] -= d * va[m];
      jb[c2] -= d * vb[m];
    }
    {
      const S d = tau[m] * seg_sum<S, P>(va[m] * ra + vb[m] * rb);
      ra -= d * va[m];
      rb -= d * vb[m];
    }
    if (row_a == m) ja[m] = beta;
    if (row_a > m) ja[m] = S(0);
    if (row_b == m) jb[m] = beta;
    if (row_b > m) jb[m] = S(0);
  }
  const S g10 = seg_sum<S, P>(va[1] * va[0] + vb[1] * vb[0]), g20 = seg_sum<S, P>(va[2] * va[0] + vb[2] * vb[0]),
          g21 = seg_sum<S, P>(va[2] * va[1] + vb[2] * vb[1]);
  const bool low_a = row_a >= 3 && valid_lane, low_b = row_b >= 3 && valid_lane;
  const S d0 = seg_sum<S, P>((low_a ? va[0] * ra : S(0)) + (low_b ? vb[0] * rb : S(0))),
          d1 = seg_sum<S, P>((low_a ? va[1] * ra : S(0)) + (low_b ? vb[1] * rb : S(0))),
          d2 = seg_sum<S, P>((low_a ? va[2] * ra : S(0)) + (low_b ? vb[2] * rb : S(0)));
  // R: rows 0 and 1 in the first lane of the landmark, row 2 in the second (a padding lane - zero - for k = 1)
  const S r00 = __shfl(ja[0], base), r01 = __shfl(ja[1], base), r02 = __shfl(ja[2], base), r11 = __shfl(jb[1], base),
          r12 = __shfl(jb[2], base), r22 = __shfl(ja[2], base + 1);
  if (r == 0 && lm_ok) {
    S* R = p.R0 + 6 * s;
    R[0] = r00;
    R[1] = r01;
    R[2] = r02;
    R[3] = r11;
    R[4] = r12;
    R[5] = r22;
    p.tauH[3 * s + 0] = tau[0];
    p.tauH[3 * s + 1] = tau[1];
    p.tauH[3 * s + 2] = tau[2];
    V4* lq = reinterpret_cast<V4*>(p.LQ + 12 * size_t(s));
    lq[0] = V4{tau[0], tau[1], tau[2], g10};
    lq[1] = V4{g20, g21, d0, d1};
    lq[2] = V4{d2, S(0), S(0), S(0)};
  }
  if (valid_lane) {
    reinterpret_cast<V4*>(p.Vh)[row] = V4{va[0], va[1], va[2], ra};
    reinterpret_cast<V4*>(p.Vh)[row + 1] = V4{vb[0], vb[1], vb[2], rb};
  }
}

template <class S>
__global__ __launch_bounds__(256) void k_s1_fused_obs(Params<S> p, ImplicitTiles it, FusedObsWaves fw) {
  stage_stamp(p.stamp);
  __shared__ __attribute__((aligned(32))) S stage[4][64 * 18];  // (32: copied out as four-scalar vectors, double4 in the double solver)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int W = blockIdx.x * 4 + wave;  // wavefront = a pair of row tiles of one class
  if (W >= fw.wave_begin[5]) return;
  if (W >= fw.wave_begin[4])
    s1_fused_obs<S, 32>(p, it.tile_begin[4], it.tile_begin[5] - it.tile_begin[4], W - fw.wave_begin[4], it.lm_begin[4],
                        it.lm_end[4], lane, stage[wave]);
  else if (W >= fw.wave_begin[3])
    s1_fused_obs<S, 16>(p, it.tile_begin[3], it.tile_begin[4] - it.tile_begin[3], W - fw.wave_begin[3], it.lm_begin[3],
                        it.lm_end[3], lane, stage[wave]);
  else if (W >= fw.wave_begin[2])
    s1_fused_obs<S, 8>(p, it.tile_begin[2], it.tile_begin[3] - it.tile_begin[2], W - fw.wave_begin[2], it.lm_begin[2],
                       it.lm_end[2], lane, stage[wave]);
  else if (W >= fw.wave_begin[1])
    s1_fused_obs<S, 4>(p, it.tile_begin[1], it.tile_begin[2] - it.tile_begin[1], W - fw.wave_begin[1], it.lm_begin[1],
                       it.lm_end[1], lane, stage[wave]);
  else
    s1_fused_obs<S, 2>(p, it.tile_begin[0], it.tile_begin[1] - it.tile_begin[0], W - fw.wave_begin[0], it.lm_begin[0],
                       it.lm_end[0], lane, stage[wave]);
}

// 32 < k <= 112: one landmark per wavefront, rows rc * 64 + lane
template <class S, int RCH>
__global__ __launch_bounds__(256) void k_s1_qr_wide(Params<S> p, int lm_begin, int lm_end) {
  using V4 = typename std::conditional<sizeof(S) == 4, float4, double4>::type;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s = lm_begin + blockIdx.x * 4 + wave;
  if (s >= lm_end) return;
  const int k = p.lm_k[s];
  const int64_t row0 = 2 * p.lm_obs[s];
  const int nrows = 2 * k;
  S jl[RCH][3], rs[RCH], vm[3][RCH], tau[3];
  bool rvalid[RCH];
#pragma unroll
  for (int rc = 0; rc < RCH; ++rc) {
    const int r = rc * 64 + lane;
    rvalid[rc] = r < nrows;
    jl[rc][0] = jl[rc][1] = jl[rc][2] = rs[rc] = S(0);
    if (rvalid[rc]) {
      const V4 v = reinterpret_cast<const V4*>(p.Vh)[row0 + r];
      jl[rc][0] = v.x;
      jl[rc][1] = v.y;
      jl[rc][2] = v.z;
      rs[rc] = v.w;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    S ss = S(0);
#pragma unroll
    for (int rc = 0; rc < RCH; ++rc) ss += jl[rc][c] * jl[rc][c];
    ss = wave_sum(ss);
    const S sc = S(1) / (p.eps + sqrt(ss));
#pragma unroll
    for (int rc = 0; rc < RCH; ++rc) jl[rc][c] *= sc;
    if (lane == 0) p.jl_scale[3 * s + c] = sc;
  }
#pragma unroll
  for (int rc = 0; rc < RCH; ++rc) {
    const int r = rc * 64 + lane;
    if (rvalid[rc]) {
      S* dst = p.JlS + 3 * (row0 + r);
      dst[0] = jl[rc][0];
      dst[1] = jl[rc][1];
      dst[2] = jl[rc][2];
      p.rS[row0 + r] = rs[rc];
    }
  }
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const S c0 = read_lane(jl[0][m], m);
    S tail = S(0);
#pragma unroll
    for (int rc = 0; rc < RCH; ++rc) {
      const int r = rc * 64 + lane;
      tail += (r > m && rvalid[rc]) ? jl[rc][m] * jl[rc][m] : S(0);
    }
    tail = wave_sum(tail);
    S beta, inv;
    if (tail <= Eps<S>::tiny) {
      tau[m] = S(0);
      beta = c0;
      inv = S(0);
    } else {
      beta = sqrt(c0 * c0 + tail);
      if (c0 >= S(0)) beta = -beta;
      inv = S(1) / (c0 - beta);
      tau[m] = (beta - c0) / beta;
    }
#pragma unroll
    for (int rc = 0; rc < RCH; ++rc) {
      const int r = rc * 64 + lane;
      vm[m][rc] = (r == m) ? S(1) : ((r > m && rvalid[rc]) ? jl[rc][m] * inv : S(0));
    }
#pragma unroll
    for (int c2 = m + 1; c2 < 3; ++c2) {
      S d = S(0);
#pragma unroll
      for (int rc = 0; rc < RCH; ++rc) d += vm[m][rc] * jl[rc][c2];
      d = tau[m] * wave_sum(d);
#pragma unroll
      for (int rc = 0; rc < RCH; ++rc) jl[rc][c2] -= d * vm[m][rc];
    }
    {
      S d = S(0);
#pragma unroll
      for (int rc = 0; rc < RCH; ++rc) d += vm[m][rc] * rs[rc];
      d = tau[m] * wave_sum(d);
#pragma unroll
      for (int rc = 0; rc < RCH; ++rc) rs[rc] -= d * vm[m][rc];
    }
#pragma unroll
    for (int rc = 0; rc < RCH; ++rc) {
      const int r = rc * 64 + lane;
      if (r == m) jl[rc][m] = beta;
      if (r > m) jl[rc][m] = S(0);
    }
  }
  S g10 = S(0), g20 = S(0), g21 = S(0), d0 = S(0), d1 = S(0), d2 = S(0);
#pragma unroll
  for (int rc = 0; rc < RCH; ++rc) {
    g10 += vm[1][rc] * vm[0][rc];
    g20 += vm[2][rc] * vm[0][rc];
    g21 += vm[2][rc] * vm[1][rc];
    const bool low = rc * 64 + lane >= 3 && rvalid[rc];
    d0 += low ? vm[0][rc] * rs[rc] : S(0);
    d1 += low ? vm[1][rc] * rs[rc] : S(0);
    d2 += low ? vm[2][rc] * rs[rc] : S(0);
  }
  g10 = wave_sum(g10);
  g20 = wave_sum(g20);
  g21 = wave_sum(g21);
  d0 = wave_sum(d0);
  d1 = wave_sum(d1);
  d2 = wave_sum(d2);
  const S r00 = read_lane(jl[0][0], 0), r01 = read_lane(jl[0][1], 0), r02 = read_lane(jl[0][2], 0),
          r11 = read_lane(jl[0][1], 1), r12 = read_lane(jl[0][2], 1), r22 = read_lane(jl[0][2], 2);
  if (lane == 0) {
    S* R = p.R0 + 6 * s;
    R[0] = r00;
    R[1] = r01;
    R[2] = r02;
    R[3] = r11;
    R[4] = r12;
    R[5] = r22;
    p.tauH[3 * s + 0] = tau[0];
    p.tauH[3 * s + 1] = tau[1];
    p.tauH[3 * s + 2] = tau[2];
    V4* lq = reinterpret_cast<V4*>(p.LQ + 12 * size_t(s));
    lq[0] = V4{tau[0], tau[1], tau[2], g10};
    lq[1] = V4{g20, g21, d0, d1};
    lq[2] = V4{d2, S(0), S(0), S(0)};
  }
#pragma unroll
  for (int rc = 0; rc < RCH; ++rc) {
    const int r = rc * 64 + lane;
    if (rvalid[rc]) reinterpret_cast<V4*>(p.Vh)[row0 + r] = V4{vm[0][rc], vm[1][rc], vm[2][rc], rs[rc]};
  }
}

// ---------------------------------------------------------------------------
// The record of damped top rows [damped Q1^T Jp D 3x9 | padding] (kTd = 32 scalars, one cache line in float) of every
// observation for the CURRENT damping, from the unscaled rows and the factors: one thread per observation (nine
// columns in registers); the three top rows are rotated straight into the damped ones by the landmark's six
// Givens rotations (k_s2_obs). Only the assembly of the reduced matrix and matrix-free E0 products read these
// records (Solver::ensure_topd). The workgroup's 128 observations are consecutive, so the records are staged in
// LDS and move as contiguous 16-byte streams.
// ---------------------------------------------------------------------------
constexpr int kS1ColsThreads = 128;
constexpr int kTdLds = 36;  // LDS stride of a staged record (32 would put every work-item's column on two banks)

template <class S>
__global__ __launch_bounds__(kS1ColsThreads) void k_s12_cols(Params<S> p, int64_t n_obs) {
  using V4 = typename std::conditional<sizeof(S) == 4, float4, double4>::type;
  using V = typename std::conditional<sizeof(S) == 4, float4, double2>::type;
  constexpr int N = 16 / int(sizeof(S)), NT = kS1ColsThreads;
  extern __shared__ __attribute__((aligned(16))) char smem_s1c[];
  S* sJ = reinterpret_cast<S*>(smem_s1c);  // [NT][18]  in: Jacobian rows, out: scaled rows
  S* sT = sJ + NT * 18;                    // [NT][kTdLds] damped Q1^T Jp (27), padded
  const int tid = threadIdx.x;
  const int64_t o_base = int64_t(blockIdx.x) * NT;
  const int n_here = int(min<int64_t>(NT, n_obs - o_base));
  const int64_t o = o_base + tid;
  const bool act = tid < n_here;
  // ---- independent loads first ----
  jp_load_rows<S>(p.JpS, p.JpT, o_base, n_here, sJ, tid, NT);
  const int64_t oc = act ? o : o_base;
  const int s = p.obs_lm[oc];
  const int cam = p.obs_cam[oc];
  const V4* __restrict__ vh = reinterpret_cast<const V4*>(p.Vh);
  const V4 va = vh[2 * oc], vb = vh[2 * oc + 1];
  S dsc[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) dsc[c] = p.pose_scaling[9 * cam + c];
  const int64_t o0 = p.lm_obs[s];
  const V4* __restrict__ lq = reinterpret_cast<const V4*>(p.LQ + 12 * size_t(s));
  const V4 q0 = lq[0], q1 = lq[1];
  const V4 w0 = vh[2 * o0], w1 = vh[2 * o0 + 1], w2 = vh[2 * o0 + 2];
  S g[16];  // the landmark's damping record: c[6], s[6], damping-row residual[3]
  {
    const V4* __restrict__ src = reinterpret_cast<const V4*>(p.givens + 16 * size_t(s));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const V4 v = src[q];
      g[4 * q] = v.x;
      g[4 * q + 1] = v.y;
      g[4 * q + 2] = v.z;
      g[4 * q + 3] = v.w;
    }
  }
  const int i = int(oc - o0);
  __syncthreads();
  if (act) {
    const S tau0 = q0.x, tau1 = q0.y, tau2 = q0.z, g10 = q0.w, g20 = q1.x, g21 = q1.y;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const S m0 = sJ[18 * tid + c] * dsc[c], m1 = sJ[18 * tid + 9 + c] * dsc[c];
      const S c0 = tau0 * (va.x * m0 + vb.x * m1);
      const S c1 = tau1 * (va.y * m0 + vb.y * m1 - c0 * g10);
      const S c2 = tau2 * (va.z * m0 + vb.z * m1 - c0 * g20 - c1 * g21);
      // rows 0..2 of Q^T Jp (column j): Q1^T Jp
      S tt[3] = {-(c0 * w0.x + c1 * w0.y + c2 * w0.z), -(c0 * w1.x + c1 * w1.y + c2 * w1.z),
                 -(c0 * w2.x + c1 * w2.y + c2 * w2.z)};
      if (i == 0) {
        tt[0] += m0;
        tt[1] += m1;
      } else if (i == 1) {
        tt[2] += m0;
      }
      // landmark damping: rotate the top rows against the three damping rows (start at zero)
      S d[3] = {S(0), S(0), S(0)};
      {
        int idx = 0;
#pragma unroll
        for (int n = 0; n < 3; ++n) {
#pragma unroll
          for (int m = 0; m <= n; ++m) {
            const S cc = g[idx], sn = g[6 + idx];
            const S x = d[n - m], y = tt[n];
            d[n - m] = cc * x + sn * y;
            tt[n] = -sn * x + cc * y;
            ++idx;
          }
        }
      }
      sT[kTdLds * tid + c] = tt[0];
      sT[kTdLds * tid + 9 + c] = tt[1];
      sT[kTdLds * tid + 18 + c] = tt[2];
    }
  }
  __syncthreads();
  // records of kTd = 32 scalars (one cache line in float) out of the 36-scalar LDS rows, as 16-byte pieces
  // (o_base is a multiple of 128: the destination is aligned; entries 27..31 are padding nobody reads)
  {
    constexpr int VPR = kTd / N;  // vectors per record
    V* dst = reinterpret_cast<V*>(p.topd + kTd * o_base);
    for (int q = tid; q < n_here * VPR; q += NT) {
      const int rec = q / VPR, v = q - VPR * rec;
      dst[q] = *reinterpret_cast<const V*>(sT + kTdLds * rec + N * v);
    }
  }
}

// ---------------------------------------------------------------------------
// Stage 2, landmark side, one thread per observation. Everything stage 2 needs per observation is LINEAR in the two
// scaled Jacobian entries (m0, m1) of a column:
//   damped top rows    tt[n] = W'[n][0] m0 + W'[n][1] m1          (n = 0..2)
//   b record           bm    = g[0] m0 + g[1] m1
// so eight coefficients (W' 3x2, g) - obtained by running the closed form of k_s12_cols on the unit inputs (1, 0) and
// (0, 1) - are all that leaves the landmark side; the camera-major pass gets them as the record WA (g and a factor
// of I - W'^T W', kernels_cam.hpp). Every work-item evaluates the six damping rotations of ITS landmark from R0,
// Q1^T r and lambda (set_landmark_damping, ipp:165-210; about 250 flops against the ~100 bytes the pass moves per
// observation - measured faster than a separate thread-per-landmark pass + a 64-byte record load, venice stage 2
// 0.443 -> 0.402 ms); the work-item of a landmark's FIRST observation also writes the landmark's records (givens,
// damped R, Q1^T r, damping-row residual, Z).
// What bounds the pass (127-133 us on venice for 0.58 GB): neither its arithmetic - the branch-free make_givens took
// 26 % of its instructions out, 1318 -> 969, without moving the time - nor the chain of dependent loads - with the
// observation's position in its landmark stored per observation the chain is two round trips instead of three: 129 ->
// 128 us, not kept - which left the ~40 sparse stores of the landmark records from one lane in five (measured
// elsewhere this round: scattered partial-line writes are what the memory system likes least): they are staged in LDS
// and leave as contiguous stores of the workgroup.
// ---------------------------------------------------------------------------
template <class S>
__global__ __launch_bounds__(256) void k_s2_obs(Params<S> p, int64_t n_obs, S lambda) {
  stage_stamp(p.stamp);
  using V4 = typename std::conditional<sizeof(S) == 4, float4, double4>::type;
  // The landmark records (rotations 16, damped triangle 6, damped Q1^T r 3, damping-row residual 3, Z 9) of the
  // landmarks whose FIRST observation lies in this workgroup - consecutive landmarks sA .. sA + nH - 1, at most 128
  // (k >= 2) - are staged here and leave as contiguous stores of the whole workgroup: written straight from the
  // first-observation work-items they were ~40 sparse stores from one lane in five, which is what bounded the pass.
  constexpr int kHeadsMax = 128, kLmRec = 37, kOffRd = 16, kOffQ = 22, kOffDr = 25, kOffZ = 28;
  __shared__ S lmrec[kHeadsMax * kLmRec];
  const int tid = threadIdx.x;
  const int64_t o_base = blockIdx.x * int64_t(256);
  const bool valid = o_base + tid < n_obs;
  const int64_t o = valid ? o_base + tid : n_obs - 1;  // (clamped: the spare lanes of the last workgroup store nothing)
  const int s = p.obs_lm[o];
  int sA, nH;
  {
    const int s_first = p.obs_lm[o_base], s_last = p.obs_lm[min(o_base + 255, n_obs - 1)];  // (workgroup-uniform)
    sA = p.lm_obs[s_first] == o_base ? s_first : s_first + 1;
    nH = s_last - sA + 1;
  }
  const V4* __restrict__ vh = reinterpret_cast<const V4*>(p.Vh);
  const V4 va = vh[2 * o], vb = vh[2 * o + 1];
  // (the tail entries of the observation's Jacobian rows, on their way into its stage-2 record: kernels_cam.hpp)
  const S jt0 = p.JpT[2 * o], jt1 = p.JpT[2 * o + 1];
  const int64_t o0 = p.lm_obs[s];
  const V4* __restrict__ lq = reinterpret_cast<const V4*>(p.LQ + 12 * size_t(s));
  const V4 q0 = lq[0], q1 = lq[1], q2 = lq[2];
  const V4 w0 = vh[2 * o0], w1 = vh[2 * o0 + 1], w2 = vh[2 * o0 + 2];
  const int i = int(o - o0);
  // ---- the landmark's damping -----------------------------------------------------------
  S T[3][4], D[3][4];
  {
    const S* R = p.R0 + 6 * size_t(s);
    T[0][0] = R[0];
    T[0][1] = R[1];
    T[0][2] = R[2];
    T[1][0] = S(0);
    T[1][1] = R[3];
    T[1][2] = R[4];
    T[2][0] = S(0);
    T[2][1] = S(0);
    T[2][2] = R[5];
    T[0][3] = w0.w;  // Q1^T r = first three entries of Q^T r (Vh[4 (2 o0 + j) + 3])
    T[1][3] = w1.w;
    T[2][3] = w2.w;
  }
  const S sl = sqrt(lambda);
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) D[a][b] = (a == b) ? sl : S(0);
  S g[16];  // c[6], s[6], damping-row residual[3], 0
  {
    int idx = 0;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
#pragma unroll
      for (int m = 0; m <= n; ++m) {
        S c = S(1), sn = S(0);
        if (lambda != S(0)) make_givens<S>(T[n][n], D[n - m][n], c, sn);
        g[idx] = c;
        g[6 + idx] = sn;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const S x = D[n - m][b], y = T[n][b];
          D[n - m][b] = c * x + sn * y;
          T[n][b] = -sn * x + c * y;
        }
        ++idx;
      }
    }
  }
  g[12] = D[0][3];
  g[13] = D[1][3];
  g[14] = D[2][3];
  g[15] = S(0);
  if (valid && i == 0) {
    S* rec = lmrec + kLmRec * (s - sA);
#pragma unroll
    for (int q = 0; q < 16; ++q) rec[q] = g[q];
    rec[kOffRd + 0] = T[0][0];
    rec[kOffRd + 1] = T[0][1];
    rec[kOffRd + 2] = T[0][2];
    rec[kOffRd + 3] = T[1][1];
    rec[kOffRd + 4] = T[1][2];
    rec[kOffRd + 5] = T[2][2];
    rec[kOffQ + 0] = T[0][3];
    rec[kOffQ + 1] = T[1][3];
    rec[kOffQ + 2] = T[2][3];
    rec[kOffDr + 0] = D[0][3];
    rec[kOffDr + 1] = D[1][3];
    rec[kOffDr + 2] = D[2][3];
    {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        S u[3] = {S(0), S(0), S(0)}, e[3] = {S(0), S(0), S(0)};
        u[j] = S(1);
        int idx = 0;
#pragma unroll
        for (int n = 0; n < 3; ++n) {
#pragma unroll
          for (int m = 0; m <= n; ++m) {
            const S x = e[n - m], y = u[n];
            e[n - m] = g[idx] * x + g[6 + idx] * y;
            u[n] = -g[6 + idx] * x + g[idx] * y;
            ++idx;
          }
        }
        u[0] = u[1] = u[2] = S(0);
#pragma unroll
        for (int n = 2; n >= 0; --n) {
#pragma unroll
          for (int m = n; m >= 0; --m) {
            --idx;
            const S x = e[n - m], y = u[n];
            e[n - m] = g[idx] * x - g[6 + idx] * y;
            u[n] = g[6 + idx] * x + g[idx] * y;
          }
        }
        rec[kOffZ + 0 + j] = u[0];
        rec[kOffZ + 3 + j] = u[1];
        rec[kOffZ + 6 + j] = u[2];
      }
    }
  }
  // ---- the observation's eight coefficients -------------------------------------------------------
  const S tau0 = q0.x, tau1 = q0.y, tau2 = q0.z, g10 = q0.w, g20 = q1.x, g21 = q1.y, d0 = q1.z, d1 = q1.w, d2 = q2.x;
  S out[2][4];  // [input][tt0 tt1 tt2 bm]
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const S m0 = e == 0 ? S(1) : S(0), m1 = e == 0 ? S(0) : S(1);
    const S c0 = tau0 * (va.x * m0 + vb.x * m1);
    const S c1 = tau1 * (va.y * m0 + vb.y * m1 - c0 * g10);
    const S c2 = tau2 * (va.z * m0 + vb.z * m1 - c0 * g20 - c1 * g21);
    S tt[3] = {-(c0 * w0.x + c1 * w0.y + c2 * w0.z), -(c0 * w1.x + c1 * w1.y + c2 * w1.z),
               -(c0 * w2.x + c1 * w2.y + c2 * w2.z)};
    S bm = -(c0 * d0 + c1 * d1 + c2 * d2);
    if (i == 0) {
      tt[0] += m0;
      tt[1] += m1;
    } else if (i == 1) {
      tt[2] += m0;
      bm += m1 * vb.w;
    } else {
      bm += m0 * va.w + m1 * vb.w;
    }
    S d[3] = {S(0), S(0), S(0)};
    int idx = 0;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
#pragma unroll
      for (int m = 0; m <= n; ++m) {
        const S cc = g[idx], sn = g[6 + idx];
        const S x = d[n - m], y = tt[n];
        d[n - m] = cc * x + sn * y;
        tt[n] = -sn * x + cc * y;
        ++idx;
      }
    }
    out[e][0] = tt[0];
    out[e][1] = tt[1];
    out[e][2] = tt[2];
    out[e][3] = bm + (d[0] * g[12] + d[1] * g[13] + d[2] * g[14]);
  }
  if (valid) {
    store_cam_record_stage2<S>(p, o, out, jt0, jt1);
    if (o >= p.w8_begin) {  // landmarks with k > 32: the two-kernel back-substitution applies W' itself
      V4* dst = reinterpret_cast<V4*>(p.W8 + 8 * (o - p.w8_begin));
      dst[0] = V4{out[0][0], out[1][0], out[0][1], out[1][1]};
      dst[1] = V4{out[0][2], out[1][2], out[0][3], out[1][3]};
    }
  }
  // ---- the staged landmark records (last: nothing of the per-observation part is live across the barrier) ----
  __syncthreads();
  {
    // contiguous copy-out: element t of an array's part belongs to record t / width, entry t % width
    for (int t = tid; t < 4 * nH; t += 256) {
      const S* r = lmrec + kLmRec * (t >> 2) + 4 * (t & 3);
      reinterpret_cast<V4*>(p.givens + 16 * size_t(sA))[t] = V4{r[0], r[1], r[2], r[3]};
    }
    for (int t = tid; t < 6 * nH; t += 256) p.Rd[6 * size_t(sA) + t] = lmrec[kLmRec * (t / 6) + kOffRd + t % 6];
    for (int t = tid; t < 3 * nH; t += 256) {
      p.q1trd[3 * size_t(sA) + t] = lmrec[kLmRec * (t / 3) + kOffQ + t % 3];
      p.damp_r[3 * size_t(sA) + t] = lmrec[kLmRec * (t / 3) + kOffDr + t % 3];
    }
    for (int t = tid; t < 9 * nH; t += 256) p.Zd[9 * size_t(sA) + t] = lmrec[kLmRec * (t / 9) + kOffZ + t % 9];
  }
}

}  // namespace rba
