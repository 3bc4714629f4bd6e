// kernels_cam.hpp — camera-major passes: the records of a camera's observations are GATHERED from landmark-major
// storage (a camera-major copy was measured and dropped, see below), float and double on the matrix cores.
//
// Rows G, M of SURVEY.md 8a: Jp_diag2 / JACOBI blocks (add_Jp_diag2, add_Jp_T_Jp_blockdiag,
// src/rootba/qr/impl/landmark_block_base.ipp:493-518, 554-569) and the SCHUR_JACOBI blocks + gradient of stage 2
// (add_Q2TJp_T_Q2TJp_blockdiag, add_Q2TJp_T_Q2Tr, ipp:520-552, 443-466; reduced over landmarks in
// linearization_qr.hpp:716-815).
//
// Data. The pass gathers, per observation of the camera (CSC index), the 72-byte record of unscaled Jacobian rows
// (JpS, stage 1; split storage - kernels.hpp, jp_row - since round 6: one aligned 64-byte line per observation) and a
// 32-byte stage-2 record WA = [ g (2) | A (2x2, row-major) | ninth entry of the two rows ] from landmark-major storage:
//   g    b record:  (Jp D)^T g is the observation's part of b
//   A    a 2x2 factor A^T A = M of the matrix M = I - W'^T W', where W' (3x2) holds the observation's rows of the
//        damped Q1 (Cholesky with the larger diagonal entry as pivot - M is singular for a landmark seen twice):
//        the kept rows of the landmark block contribute
//            Jp^T (Q2 Q2^T)_oo Jp = Jp^T M Jp = (A Jp)^T (A Jp)
//        to the camera's diagonal block.
// (Tried in round 3 and dropped: a CAMERA-MAJOR copy of both records, so that this pass streams instead of gathering.
//  The pass itself went from 237-287 to 215 us, but the scattered 72- / 32-byte record writes cost the geometry pass
//  +160 us and the landmark side of stage 2 +70 us on venice - partial cache lines written from many workgroups,
//  XCD-contiguous block order or not - a net loss of 0.18 ms per iteration: profiles/r3_camera_major_copy_kernel_stats.csv.)
//
// Numerics. Round 2 formed the SCHUR_JACOBI block as D (G - T) D with G = sum Jp^T Jp and T = sum (W'Jp)^T (W'Jp)
// accumulated separately in float: the subtraction cancels and carries the ACCUMULATION rounding of both sums
// (~ eps sqrt(n) |G|) into a result that can be much smaller than G - the root cause of the assembled operator
// losing definiteness on final-13682. Here the difference is taken PER OBSERVATION in the
// 2x2 matrix M (entries of order 1, error eps), factored, and only positive semi-definite rank-2 terms are summed:
// no cancellation between sums, the block is symmetric PSD by construction, and the matrix cores need ONE
// v_mfma_f32_16x16x4_f32 per TWO observations (rank 2 x 2 = K 4) instead of one per observation for T plus one per
// two for G.
#pragma once

#include "kernels.hpp"

namespace rba {

constexpr int kRecW = 8;     // scalars per WA record
constexpr int kRecG = 0;     // offset of g in a WA record
constexpr int kRecA = 2;     // offset of A

// MODE 0: stage-2 pass. blocks = D K D + lambda I (SCHUR_JACOBI) or D G D + lambda I (JACOBI / power series; then
//         sdiag = D K D when the assembled matrix wants its diagonal), b = D t.
//         GRAM = 1 (first stage 2 of a linearisation point on one GPU): Jp_diag2 and the pose scaling D are formed
//         here too (and B_mid = D G D when the preconditioner needs it), from the same staged records.
// MODE 1: stage-1 Gram pass on its own (sharded runs: Jp_diag2 is all-reduced before D exists; callers that read
//         Jp_diag2 right after rba_linearize): Jp_diag2 and the UNSCALED G into B_mid (k_scale_gram scales it).
template <class S, int MODE>
__global__ __launch_bounds__(256) void k_cam_pass_mfma(Params<S> p, S lambda, int GRAM) {
  using M = Mfma<S>;
  using V2 = typename M::V2;
  using V4 = typename M::V4;
  using Acc = typename M::acc;
  constexpr int CH = kCamChunk, RW = 26;  // staged record: [Jp 18 | g 2 | A 4 | (2 unused)]
  __shared__ __attribute__((aligned(16))) S stage[4][CH * RW + 6];
  __shared__ S tile[4][16][16];
  __shared__ double bsum[4][7][9];
  __shared__ double dsum[4][7][9];
  __shared__ S dsc[9];
  const int c = xcd_swizzled_camera(p.n_cams);
  if (c >= p.n_cams) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t t0 = p.cam_obs_off[c], t1 = p.cam_obs_off[c + 1];
  const bool want_K = MODE == 0 && (!p.jacobi || p.want_sdiag);
  const bool want_G = MODE == 1 || (GRAM && p.jacobi);
  const bool want_d = MODE == 1 || GRAM;
  Acc accK = {0, 0, 0, 0}, accK2 = {0, 0, 0, 0}, accG = {0, 0, 0, 0};
  const int i = lane & 15, kk = lane >> 4, i9 = min(i, 8);
  const int g = lane / 9, a = lane - 9 * g;
  double accb = 0, accd = 0;
  S* lds = stage[wave];
  // (the observation indices of the wave's next chunk are requested before the current one is processed; clamped, not
  //  predicated: every lane holds a valid observation of this camera)
  int idxreg = t1 > t0 ? p.cam_obs[min<int64_t>(t0 + CH * wave + lane, t1 - 1)] : 0;
  for (int64_t base = t0 + CH * wave; base < t1; base += 4 * CH) {
    const int cnt = int(min<int64_t>(CH, t1 - base));
    const int idxnext = p.cam_obs[min<int64_t>(base + 4 * CH + lane, t1 - 1)];
    // Jacobian rows: nine two-scalar pieces per record; WA: two four-scalar pieces (kept at two-scalar granularity
    // in LDS: the 26-scalar record stride is not a multiple of four). Every load of the chunk is issued before the first
    // LDS store and none is conditional (lanes past the chunk re-read one of its first records: `idxreg` is a valid
    // observation in every lane) - a load inside `if (q < ...)` is a basic block of its own that waits for its data
    // before the next one is issued: seven memory round trips per chunk instead of two.
    // (Measured in round 6 and not kept: the record loads of chunk k + 1 issued ahead of the staging and multiplication
    //  of chunk k - 153 us against 150: the pass is not bound by the round trips of one wavefront.)
    // (split storage of the rows, kernels.hpp: the main part of an observation is ONE aligned line - four 16-byte
    //  pieces in float - and its two tail entries ride in the stage-2 record, WA[6..7]; the Gram pass on its own, which
    //  has no stage-2 record yet, reads them from JpT)
    constexpr int NJ = CH * 4 / 64;
    static_assert(CH * 4 % 64 == 0 && CH <= 32, "whole passes of 16-byte pieces; one record half per lane");
    V4 jv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int q = j * 64 + lane;
      const int o = __shfl(idxreg, (q >> 2) & 31);
      jv[j] = reinterpret_cast<const V4*>(p.JpS)[int64_t(o) * 4 + (q & 3)];
    }
    V4 w = {0, 0, 0, 0};
    V2 jt = {0, 0};
    if (MODE == 0) {
      const int o = __shfl(idxreg, (lane >> 1) & 31);
      w = *reinterpret_cast<const V4*>(p.WA + int64_t(o) * kRecW + 4 * (lane & 1));
    } else {
      jt = reinterpret_cast<const V2*>(p.JpT)[__shfl(idxreg, lane & 31)];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int q = j * 64 + lane;
      const int r = q >> 2, pc = q & 3;
      if (r < cnt) {
        S* d = lds + r * RW + 9 * (pc >> 1) + 4 * (pc & 1);
        d[0] = jv[j].x, d[1] = jv[j].y, d[2] = jv[j].z, d[3] = jv[j].w;
      }
    }
    if (MODE == 0) {
      const int r = lane >> 1, h = lane & 1;
      if (r < cnt) {
        S* d = lds + r * RW + 18 + 4 * h;
        *reinterpret_cast<V2*>(d) = V2{w.x, w.y};
        if (h == 0) {
          *reinterpret_cast<V2*>(d + 2) = V2{w.z, w.w};
        } else {  // second half of the record: [a10 a11 | tail of row 0, tail of row 1]
          lds[r * RW + 8] = w.z;
          lds[r * RW + 17] = w.w;
        }
      }
    } else if (lane < cnt) {
      lds[lane * RW + 8] = jt.x;
      lds[lane * RW + 17] = jt.y;
    }
    wave_lds_fence();
    if (want_K) {
      // two observations = four rows of Y = A Jp per instruction; two accumulators (no dependent chain)
      for (int s = 0; s < cnt; s += 4) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int so = s + 2 * h + (kk >> 1);
          // (unpredicated: the tile's rows 9 .. 15 repeat column 8 and are never read back; records past the chunk's
          //  end - stale LDS - are replaced by zeros with a select instead of a branch around the LDS reads)
          const S* rec = lds + so * RW;
          const S* arow = rec + 18 + kRecA + 2 * (kk & 1);
          S v = fma(arow[0], rec[i9], M::mul_rn(arow[1], rec[9 + i9]));
          v = so < cnt ? v : S(0);
          if (h == 0)
            accK = M::mma(v, v, accK);
          else
            accK2 = M::mma(v, v, accK2);
        }
      }
    }
    if (want_G) {
      for (int s = 0; s < cnt; s += 2) {
        const int so = s + (kk >> 1);
        const S vg = lds[so * RW + 9 * (kk & 1) + i9];
        const S v = so < cnt ? vg : S(0);
        accG = M::mma(v, v, accG);
      }
    }
    if (lane < 63)
      for (int r = g; r < cnt; r += 7) {
        const S* rec = lds + r * RW;
        if (MODE == 0) accb += double(fma(rec[a], rec[18 + kRecG], M::mul_rn(rec[9 + a], rec[18 + kRecG + 1])));
        if (want_d) accd += double(fma(rec[a], rec[a], M::mul_rn(rec[9 + a], rec[9 + a])));
      }
    wave_lds_fence();  // the next chunk overwrites the staging buffer
    idxreg = idxnext;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) accK[r] += accK2[r];
  if (lane < 63) {
    bsum[wave][g][a] = accb;
    dsum[wave][g][a] = accd;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[wave][M::row(lane, r)][lane & 15] = accG[r];
  __syncthreads();
  S gsum = S(0);
  if (want_G && tid < 81) {
    const int ii = tid / 9, jj = tid - 9 * ii;
#pragma unroll
    for (int w = 0; w < 4; ++w) gsum += tile[w][ii][jj];
  }
  if (want_d && tid >= 128 && tid < 137) {
    const int aa = tid - 128;
    double sum = 0;
    for (int w = 0; w < 4; ++w)
      for (int gg = 0; gg < 7; ++gg) sum += dsum[w][gg][aa];
    const S d2 = S(sum);
    p.jp_diag2[9 * c + aa] = d2;
    if (MODE == 0) {
      const S sc = S(1) / (p.eps + sqrt(d2));  // k_pose_scaling
      p.pose_scaling[9 * c + aa] = sc;
      dsc[aa] = sc;
    }
  }
  if (MODE == 1) {
    if (tid < 81) p.B_mid[81 * c + tid] = gsum;
    return;
  }
  if (!want_d && tid < 9) dsc[tid] = p.pose_scaling[9 * c + tid];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[wave][M::row(lane, r)][lane & 15] = accK[r];
  __syncthreads();
  if (tid < 81) {
    const int ii = tid / 9, jj = tid - 9 * ii;
    S t = S(0);
#pragma unroll
    for (int w = 0; w < 4; ++w) t += tile[w][ii][jj];
    const S dd = dsc[ii] * dsc[jj];
    const S kd = M::mul_rn(t, dd);  // D K D: the camera's diagonal block of the reduced matrix (lambda = 0)
    if (p.jacobi) {
      S bm;
      if (GRAM) {
        bm = M::mul_rn(gsum, dd);
        p.B_mid[81 * c + tid] = bm;
      } else {
        bm = p.B_mid[81 * c + tid];
      }
      p.blocks[81 * c + tid] = bm + (ii == jj ? lambda : S(0));
      if (p.want_sdiag) p.sdiag[81 * c + tid] = kd;
    } else {
      p.blocks[81 * c + tid] = kd + (ii == jj ? lambda : S(0));
    }
  }
  if (tid >= 128 && tid < 137) {
    const int aa = tid - 128;
    double sum = 0.0;
    for (int w = 0; w < 4; ++w)
      for (int gg = 0; gg < 7; ++gg) sum += bsum[w][gg][aa];
    p.b[9 * c + aa] = S(sum * double(dsc[aa]));
  }
}

// an observation's stage-2 record WA from its eight stage-2 coefficients
// W' (3x2: w[n][e] = out[e][n]) and g (out[e][3]): g, and a factor A^T A = M = I - W'^T W' (PSD up to rounding:
// the columns of W' are three entries each of two columns of an orthogonal matrix). Cholesky with the LARGER
// diagonal entry as pivot: M is (numerically) singular for a landmark with two observations, and dividing by the
// small pivot would amplify its rounding error eps / m into the other diagonal entry; negative remainders clamp to 0.
template <class S>
__device__ __forceinline__ void store_cam_record_stage2(const Params<S>& p, int64_t o, const S out[2][4], S jt0, S jt1) {
  using V4 = typename std::conditional<sizeof(S) == 4, float4, double4>::type;
  const S m00 = S(1) - (out[0][0] * out[0][0] + out[0][1] * out[0][1] + out[0][2] * out[0][2]);
  const S m01 = -(out[0][0] * out[1][0] + out[0][1] * out[1][1] + out[0][2] * out[1][2]);
  const S m11 = S(1) - (out[1][0] * out[1][0] + out[1][1] * out[1][1] + out[1][2] * out[1][2]);
  S a00, a01, a10, a11;
  if (m00 >= m11) {  // M = L L^T, A = L^T
    a00 = sqrt(max(m00, S(0)));
    a01 = a00 > S(0) ? m01 / a00 : S(0);
    a10 = S(0);
    a11 = sqrt(max(m11 - a01 * a01, S(0)));
  } else {  // M = U U^T, A = U^T   (same guards: at lambda = 0 both diagonal entries of a landmark seen twice can
            //                      round to <= 0; max() also maps a NaN m11 to 0)
    a11 = sqrt(max(m11, S(0)));
    a10 = a11 > S(0) ? m01 / a11 : S(0);
    a01 = S(0);
    a00 = sqrt(max(m00 - a10 * a10, S(0)));
  }
  // (+ jt0, jt1: the ninth entry of the observation's two Jacobian rows, JpT[2 o], JpT[2 o + 1] - read by the caller
  //  with its first loads: the camera-major pass then needs no third line per observation)
  V4* rec = reinterpret_cast<V4*>(p.WA + o * kRecW);
  rec[0] = V4{out[0][3], out[1][3], a00, a01};
  rec[1] = V4{a10, a11, jt0, jt1};
}

}  // namespace rba
