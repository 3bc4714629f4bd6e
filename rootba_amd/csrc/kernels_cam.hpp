// kernels_cam.hpp — camera-major passes over the CAMERA-MAJOR record copy (round 3).
//
// Rows G, M of SURVEY.md 8a: Jp_diag2 / JACOBI blocks (add_Jp_diag2, add_Jp_T_Jp_blockdiag,
// src/rootba/qr/impl/landmark_block_base.ipp:493-518, 554-569) and the SCHUR_JACOBI blocks + gradient of stage 2
// (add_Q2TJp_T_Q2TJp_blockdiag, add_Q2TJp_T_Q2Tr, ipp:520-552, 443-466; reduced over landmarks in
// linearization_qr.hpp:716-815).
//
// Data. The pass gathers, per observation of the camera (CSC index), the 72-byte record of unscaled Jacobian rows
// (JpS, stage 1) and a 32-byte stage-2 record WA = [ g (2) | A (2x2, row-major) | 0 0 ] from landmark-major storage:
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
template <int MODE>
__global__ __launch_bounds__(256) void k_cam_pass_mfma(Params<float> p, float lambda, int GRAM) {
  constexpr int CH = kCamChunk, RW = 26;  // staged record: [Jp 18 | g 2 | A 4 | (2 unused)]
  __shared__ __attribute__((aligned(16))) float stage[4][CH * RW + 6];
  __shared__ float tile[4][16][16];
  __shared__ double bsum[4][7][9];
  __shared__ double dsum[4][7][9];
  __shared__ float dsc[9];
  const int c = xcd_swizzled_camera(p.n_cams);
  if (c >= p.n_cams) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t t0 = p.cam_obs_off[c], t1 = p.cam_obs_off[c + 1];
  const bool want_K = MODE == 0 && (!p.jacobi || p.want_sdiag);
  const bool want_G = MODE == 1 || (GRAM && p.jacobi);
  const bool want_d = MODE == 1 || GRAM;
  f32x4 accK = {0.f, 0.f, 0.f, 0.f}, accK2 = {0.f, 0.f, 0.f, 0.f}, accG = {0.f, 0.f, 0.f, 0.f};
  const int i = lane & 15, kk = lane >> 4;
  const int g = lane / 9, a = lane - 9 * g;
  double accb = 0, accd = 0;
  float* lds = stage[wave];
  // (the observation indices of the wave's next chunk are requested before the current one is processed; clamped, not
  //  predicated: every lane holds a valid observation of this camera)
  int idxreg = t1 > t0 ? p.cam_obs[min<int64_t>(t0 + CH * wave + lane, t1 - 1)] : 0;
  for (int64_t base = t0 + CH * wave; base < t1; base += 4 * CH) {
    const int cnt = int(min<int64_t>(CH, t1 - base));
    const int idxnext = p.cam_obs[min<int64_t>(base + 4 * CH + lane, t1 - 1)];
    // Jacobian rows: nine 8-byte pieces per record; WA: two 16-byte pieces (kept at 8-byte granularity in LDS:
    // the 26-float record stride is not a multiple of 16 bytes). Every load of the chunk is issued before the first
    // LDS store and none is conditional (lanes past the chunk re-read one of its first records: `idxreg` is a valid
    // observation in every lane) - a load inside `if (q < ...)` is a basic block of its own that waits for its data
    // before the next one is issued: seven memory round trips per chunk instead of two.
    constexpr int NJ = (CH * 9 + 63) / 64;
    float2 jv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int q = j * 64 + lane;
      const int r = q / 9, pc = q - 9 * r;
      const int o = __shfl(idxreg, r & 31);
      jv[j] = *reinterpret_cast<const float2*>(p.JpS + int64_t(o) * 18 + 2 * pc);
    }
    float4 w = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
      const int o = __shfl(idxreg, (lane >> 1) & 31);
      w = *reinterpret_cast<const float4*>(p.WA + int64_t(o) * kRecW + 4 * (lane & 1));
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int q = j * 64 + lane;
      const int r = q / 9, pc = q - 9 * r;
      if (q < cnt * 9) *reinterpret_cast<float2*>(lds + r * RW + 2 * pc) = jv[j];
    }
    if (MODE == 0) {
      const int r = lane >> 1, h = lane & 1;
      if (r < cnt) {
        float* d = lds + r * RW + 18 + 4 * h;
        *reinterpret_cast<float2*>(d) = float2{w.x, w.y};
        *reinterpret_cast<float2*>(d + 2) = float2{w.z, w.w};
      }
    }
    wave_lds_fence();
    if (want_K) {
      // two observations = four rows of Y = A Jp per instruction; two accumulators (no dependent chain)
      for (int s = 0; s < cnt; s += 4) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int so = s + 2 * h + (kk >> 1);
          float v = 0.f;
          if (i < 9 && so < cnt) {
            const float* rec = lds + so * RW;
            const float* arow = rec + 18 + kRecA + 2 * (kk & 1);
            v = fmaf(arow[0], rec[i], __fmul_rn(arow[1], rec[9 + i]));
          }
          if (h == 0)
            accK = __builtin_amdgcn_mfma_f32_16x16x4f32(v, v, accK, 0, 0, 0);
          else
            accK2 = __builtin_amdgcn_mfma_f32_16x16x4f32(v, v, accK2, 0, 0, 0);
        }
      }
    }
    if (want_G) {
      for (int s = 0; s < cnt; s += 2) {
        const int so = s + (kk >> 1);
        const float v = (i < 9 && so < cnt) ? lds[so * RW + 9 * (kk & 1) + i] : 0.f;
        accG = __builtin_amdgcn_mfma_f32_16x16x4f32(v, v, accG, 0, 0, 0);
      }
    }
    if (lane < 63)
      for (int r = g; r < cnt; r += 7) {
        const float* rec = lds + r * RW;
        if (MODE == 0) accb += double(fmaf(rec[a], rec[18 + kRecG], __fmul_rn(rec[9 + a], rec[18 + kRecG + 1])));
        if (want_d) accd += double(fmaf(rec[a], rec[a], __fmul_rn(rec[9 + a], rec[9 + a])));
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
  for (int r = 0; r < 4; ++r) tile[wave][(lane >> 4) * 4 + r][lane & 15] = accG[r];
  __syncthreads();
  float gsum = 0.f;
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
    const float d2 = float(sum);
    p.jp_diag2[9 * c + aa] = d2;
    if (MODE == 0) {
      const float sc = 1.f / (p.eps + sqrtf(d2));  // k_pose_scaling
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
  for (int r = 0; r < 4; ++r) tile[wave][(lane >> 4) * 4 + r][lane & 15] = accK[r];
  __syncthreads();
  if (tid < 81) {
    const int ii = tid / 9, jj = tid - 9 * ii;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) t += tile[w][ii][jj];
    const float dd = dsc[ii] * dsc[jj];
    const float kd = __fmul_rn(t, dd);  // D K D: the camera's diagonal block of the reduced matrix (lambda = 0)
    if (p.jacobi) {
      float bm;
      if (GRAM) {
        bm = __fmul_rn(gsum, dd);
        p.B_mid[81 * c + tid] = bm;
      } else {
        bm = p.B_mid[81 * c + tid];
      }
      p.blocks[81 * c + tid] = bm + (ii == jj ? lambda : 0.f);
      if (p.want_sdiag) p.sdiag[81 * c + tid] = kd;
    } else {
      p.blocks[81 * c + tid] = kd + (ii == jj ? lambda : 0.f);
    }
  }
  if (tid >= 128 && tid < 137) {
    const int aa = tid - 128;
    double sum = 0.0;
    for (int w = 0; w < 4; ++w)
      for (int gg = 0; gg < 7; ++gg) sum += bsum[w][gg][aa];
    p.b[9 * c + aa] = float(sum * double(dsc[aa]));
  }
}

// generic (double): the same pass on the vector ALU with double accumulators, 64 records staged per workgroup
template <class S, int MODE>
__global__ __launch_bounds__(256) void k_cam_pass(Params<S> p, S lambda, int GRAM) {
  constexpr int TILE = 64, RW = 18 + kRecW;
  __shared__ S rec[TILE][RW];  // [Jp 18 | g 2 | A 4 | 0 0]
  __shared__ int olist[TILE];
  __shared__ double red[3][81], redG[3][81];
  __shared__ double dsc[9];
  const int c = blockIdx.x;
  const int tid = threadIdx.x;
  const int grp = tid / 81, e = tid - 81 * grp, ea = e / 9, eb = e - 9 * ea;
  const bool want_K = MODE == 0 && (!p.jacobi || p.want_sdiag);
  const bool want_G = MODE == 1 || GRAM;  // (its diagonal is Jp_diag2)
  double acc = 0, accG = 0;
  const int64_t t0 = p.cam_obs_off[c], t1 = p.cam_obs_off[c + 1];
  for (int64_t base = t0; base < t1; base += TILE) {
    const int n = int(min<int64_t>(TILE, t1 - base));
    __syncthreads();
    if (tid < n) olist[tid] = p.cam_obs[base + tid];
    __syncthreads();
    for (int idx = tid; idx < n * RW; idx += 256) {
      const int q = idx / RW, f = idx - RW * q;
      if (f < 18)
        rec[q][f] = p.JpS[int64_t(olist[q]) * 18 + f];
      else if (MODE == 0)
        rec[q][f] = p.WA[int64_t(olist[q]) * kRecW + (f - 18)];
    }
    __syncthreads();
    if (grp < 3) {
      for (int q = grp; q < n; q += 3) {
        const S* r = rec[q];
        if (want_K) {
          const S a00 = r[18 + kRecA], a01 = r[18 + kRecA + 1], a10 = r[18 + kRecA + 2], a11 = r[18 + kRecA + 3];
          const S ya = a00 * r[ea] + a01 * r[9 + ea], yb = a00 * r[eb] + a01 * r[9 + eb];
          const S za = a10 * r[ea] + a11 * r[9 + ea], zb = a10 * r[eb] + a11 * r[9 + eb];
          acc += double(ya * yb + za * zb);
        }
        if (want_G) accG += double(r[ea] * r[eb] + r[9 + ea] * r[9 + eb]);
      }
    } else if (MODE == 0 && tid < 252) {
      const int a = tid - 243;
      for (int q = 0; q < n; ++q) acc += double(rec[q][a] * rec[q][18 + kRecG] + rec[q][9 + a] * rec[q][18 + kRecG + 1]);
    }
  }
  if (grp < 3) {
    red[grp][e] = acc;
    redG[grp][e] = accG;
  }
  __syncthreads();
  double gsum = 0;
  if (want_G && tid < 81) {
    gsum = redG[0][tid] + redG[1][tid] + redG[2][tid];
    if (ea == eb) {
      const S d2 = S(gsum);
      p.jp_diag2[9 * c + ea] = d2;
      if (MODE == 0) {
        const S sc = S(1) / (p.eps + sqrt(d2));
        p.pose_scaling[9 * c + ea] = sc;
        dsc[ea] = double(sc);
      }
    }
  }
  if (MODE == 1) {
    if (tid < 81) p.B_mid[81 * c + tid] = S(gsum);
    return;
  }
  if (!want_G && tid < 9) dsc[tid] = double(p.pose_scaling[9 * c + tid]);
  __syncthreads();
  if (tid < 81) {
    const double dd = dsc[ea] * dsc[eb];
    const double kd = (red[0][tid] + red[1][tid] + red[2][tid]) * dd;  // D K D
    if (p.jacobi) {
      double bm;
      if (GRAM) {
        bm = double(S(gsum) * S(dd));  // as k_scale_gram applied to the stored Gram block
        p.B_mid[81 * c + tid] = S(bm);
      } else {
        bm = double(p.B_mid[81 * c + tid]);
      }
      p.blocks[81 * c + tid] = S(bm + (ea == eb ? double(lambda) : 0.0));
      if (p.want_sdiag) p.sdiag[81 * c + tid] = S(kd);
    } else {
      p.blocks[81 * c + tid] = S(kd + (ea == eb ? double(lambda) : 0.0));
    }
  }
  if (tid >= 243 && tid < 252) p.b[9 * c + (tid - 243)] = S(acc * dsc[tid - 243]);
}

// an observation's stage-2 record WA from its eight stage-2 coefficients
// W' (3x2: w[n][e] = out[e][n]) and g (out[e][3]): g, and a factor A^T A = M = I - W'^T W' (PSD up to rounding:
// the columns of W' are three entries each of two columns of an orthogonal matrix). Cholesky with the LARGER
// diagonal entry as pivot: M is (numerically) singular for a landmark with two observations, and dividing by the
// small pivot would amplify its rounding error eps / m into the other diagonal entry; negative remainders clamp to 0.
template <class S>
__device__ __forceinline__ void store_cam_record_stage2(const Params<S>& p, int64_t o, const S out[2][4]) {
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
  } else {  // M = U U^T, A = U^T
    a11 = sqrt(m11);
    a10 = m01 / a11;
    a01 = S(0);
    a00 = sqrt(max(m00 - a10 * a10, S(0)));
  }
  V4* rec = reinterpret_cast<V4*>(p.WA + o * kRecW);
  rec[0] = V4{out[0][3], out[1][3], a00, a01};
  rec[1] = V4{a10, a11, S(0), S(0)};
}

}  // namespace rba
