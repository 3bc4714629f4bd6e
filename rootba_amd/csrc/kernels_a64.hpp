// kernels_a64.hpp — the ASSEMBLED reduced camera matrix of a FLOAT solver, evaluated in DOUBLE from the float factors.
//
// Why. Long PCG solves run on the assembled matrix S = sum_l A_l^T A_l instead of the matrix-free product
// (solver.hip: pcg()). Rounds 2-3 assembled and stored S in float: S~ = S + E with |E| ~ eps |S|, and
// the curvature p^T S~ p of a near-null direction then carries a relative error eps * kappa(S) where the square-root
// product |A p|^2 (orthogonal transformations applied to J p, the reason the reference's solver exists,
// src/rootba/qr/linearization_qr.hpp:406-429) carries eps * sqrt(kappa): increments of 150-500-iteration solves were
// 3-10 x further from the float64 iterate than the float32 reference's. ANY float storage of
// S has that error, however the entries are computed. So the matrix is DOUBLE - and it is computed in double from
// the float factors in a way that keeps it the exact (to 1e-16) reduced camera matrix of a nearby (eps_float
// backward error in J, the class of the matrix-free float product) problem:
//
//   * reflectors: the float vectors v_m are kept, tau_m is RE-DERIVED as 2 / (v_m^T v_m) in double, which makes every
//     H_m = I - tau_m v_m v_m^T orthogonal to double precision (the float tau makes it orthogonal to 6e-8 only);
//     the products v_a^T v_b of the compact application are re-summed in double as well (k_a64_landmark);
//   * landmark damping: the six rotations are re-evaluated in double from the float triangle R0 and lambda
//     (c^2 + s^2 = 1 to 1e-16), so the 3 x 2 coefficient block W' of an observation (its rows of the damped Q1) is a
//     piece of an orthogonal matrix to double precision;
//   * with that, the identities the float assembly relied on hold to 1e-16 instead of 6e-8:
//        diagonal block      D [sum_o (A_o Jp_o)^T (A_o Jp_o)] D,   A_o^T A_o = I - W'_o^T W'_o      (k_a64_diag)
//        off-diagonal block  - D_c [sum_l (W'_i Jp_i)^T (W'_j Jp_j)] D_d                               (k_a64_offdiag)
//     The Jacobian rows Jp (float, exact inputs) enter as they are and the Jacobi scaling D (float) multiplies the
//     finished blocks, in double: diagonal and off-diagonal blocks are sums over the SAME rows (a record of rows
//     scaled and rounded to float would be a different Jacobian in the off-diagonal blocks than on the diagonal - an
//     eps_float inconsistency of the matrix, the error class this file exists to avoid).
//   * the per-observation record of the assembly is ONE 128-byte cache line: [Jp 18 float | W' 6 double | pad]; the
//     pair gather of the off-diagonal blocks forms the three damped top rows W' Jp on the fly (two FMAs per operand
//     of the matrix-core instruction) instead of fetching a 256-byte record of them.
// The PCG vectors stay float, as in the float reference; k_pcgs_spmv multiplies double blocks with float operands
// in double (kernels_pcg.hpp). Cost on venice-1778: see DESIGN.md 3c.
#pragma once

#include "kernels_cam.hpp"
#include "kernels_sc.hpp"

namespace rba {

struct A64Params {
  int n_cams, n_lms;
  const int* __restrict__ lm_k;
  const int64_t* __restrict__ lm_obs;
  const int* __restrict__ obs_cam;
  const int* __restrict__ obs_lm;
  const int64_t* __restrict__ cam_obs_off;
  const int* __restrict__ cam_obs;
  const float* __restrict__ JpS;           // [2 n_obs][8]  rows, entries 0..7 (split storage: kernels.hpp, jp_row)
  const float* __restrict__ JpT;           // [2 n_obs]     entry 8
  const float* __restrict__ Vh;            // [2 n_obs][4]
  const float* __restrict__ tauH;          // [3 n_lms] (only its zeros are used: a skipped reflector stays skipped)
  const float* __restrict__ R0;            // [6 n_lms]
  const float* __restrict__ pose_scaling;  // [9 n_cams]
  double* LQ;    // [n_lms][8]  tau0 tau1 tau2 g10 g20 g21 - -   (double re-derivation of Params::LQ)
  double* A;     // [n_obs][4]  2x2 factor A, A^T A = I - W'^T W'
  double* rec;   // [n_obs][16] one cache line per observation: Jp (18 float = 9 double slots) | W' (3x2 double, row-major) | pad
};

constexpr int kA64Lq = 8;
constexpr int kA64Rec = 16;   // doubles per record
constexpr int kA64RecW = 9;   // offset of W' (doubles)

// the six sums of one landmark's reflector rows -> tau (double), cross products
__device__ __forceinline__ void a64_store_lq(const A64Params& p, int s, const double n[3], double g10, double g20,
                                             double g21) {
  double* lq = p.LQ + size_t(kA64Lq) * s;
#pragma unroll
  for (int m = 0; m < 3; ++m) lq[m] = p.tauH[3 * s + m] == 0.0f ? 0.0 : 2.0 / n[m];
  lq[3] = g10;
  lq[4] = g20;
  lq[5] = g21;
  lq[6] = lq[7] = 0.0;
}

// short tracks (k <= 32): one lane per block row on the wave tiles of the QR pass (kernels_s1.hpp: a landmark = an aligned
// group of P2 lanes), six segmented sums in double. (A work-item per landmark walking its rows fetched every cache line
// four times - 686 MB for 160 MB of rows, 150 us: profiles/r4_pmc_stage_traffic.csv, first half of round 4.)
template <int P2>
__device__ __forceinline__ void a64_landmark_tile(const A64Params& p, const int2* __restrict__ OT, size_t T, int t_in_class,
                                                  int lm_begin, int lm_end, int lane) {
  constexpr int LPW = 64 / P2;
  const int seg = lane / P2, r = lane - P2 * seg;
  const int s = lm_begin + t_in_class * LPW + seg;
  const int2 slot = OT[T * 32 + (lane >> 1)];  // {camera, first block row} of the lane's observation
  const int64_t row = slot.x >= 0 ? slot.y + (lane & 1) : -1;  // -1: padding lane
  double v0 = 0, v1 = 0, v2 = 0;
  if (row >= 0) {
    const float4 v = reinterpret_cast<const float4*>(p.Vh)[row];
    v0 = v.x;
    v1 = v.y;
    v2 = v.z;
  }
  double n[3] = {seg_sum<double, P2>(v0 * v0), seg_sum<double, P2>(v1 * v1), seg_sum<double, P2>(v2 * v2)};
  const double g10 = seg_sum<double, P2>(v1 * v0), g20 = seg_sum<double, P2>(v2 * v0), g21 = seg_sum<double, P2>(v2 * v1);
  if (r == 0 && s < lm_end) a64_store_lq(p, s, n, g10, g20, g21);
}

__global__ __launch_bounds__(256) void k_a64_landmark(A64Params p, const int2* __restrict__ OT, ImplicitTiles it) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int T = blockIdx.x * 4 + wave;
  if (T >= it.tile_begin[5]) return;
  if (T >= it.tile_begin[4])
    a64_landmark_tile<64>(p, OT, T, T - it.tile_begin[4], it.lm_begin[4], it.lm_end[4], lane);
  else if (T >= it.tile_begin[3])
    a64_landmark_tile<32>(p, OT, T, T - it.tile_begin[3], it.lm_begin[3], it.lm_end[3], lane);
  else if (T >= it.tile_begin[2])
    a64_landmark_tile<16>(p, OT, T, T - it.tile_begin[2], it.lm_begin[2], it.lm_end[2], lane);
  else if (T >= it.tile_begin[1])
    a64_landmark_tile<8>(p, OT, T, T - it.tile_begin[1], it.lm_begin[1], it.lm_end[1], lane);
  else
    a64_landmark_tile<4>(p, OT, T, T - it.tile_begin[0], it.lm_begin[0], it.lm_end[0], lane);
}

// long tracks: one wavefront per landmark
__global__ __launch_bounds__(256) void k_a64_landmark_wave(A64Params p, int lm_begin, int lm_end) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s = lm_begin + blockIdx.x * 4 + wave;
  if (s >= lm_end) return;
  const int nrows = 2 * p.lm_k[s];
  const float4* __restrict__ vh = reinterpret_cast<const float4*>(p.Vh) + 2 * p.lm_obs[s];
  double n[3] = {0, 0, 0}, g10 = 0, g20 = 0, g21 = 0;
  for (int r = lane; r < nrows; r += 64) {
    const float4 v = vh[r];
    const double v0 = v.x, v1 = v.y, v2 = v.z;
    n[0] += v0 * v0;
    n[1] += v1 * v1;
    n[2] += v2 * v2;
    g10 += v1 * v0;
    g20 += v2 * v0;
    g21 += v2 * v1;
  }
#pragma unroll
  for (int m = 0; m < 3; ++m) n[m] = wave_sum(n[m]);
  g10 = wave_sum(g10);
  g20 = wave_sum(g20);
  g21 = wave_sum(g21);
  if (lane == 0) a64_store_lq(p, s, n, g10, g20, g21);
}

// One work-item per observation (the double twin of k_s2_obs, without the b part): the landmark's six damping
// rotations, the observation's coefficient block W' (3 x 2), the factor A of I - W'^T W', and the observation's record
// [Jp | W']. The workgroup's observations are consecutive: Jacobian rows in and records out move as contiguous 16-byte
// streams through LDS.
constexpr int kA64Threads = 128;

__global__ __launch_bounds__(kA64Threads) void k_a64_obs(A64Params p, int64_t n_obs, double lambda) {
  constexpr int NT = kA64Threads;
  __shared__ __attribute__((aligned(16))) double sRec[NT * kA64Rec];
  float* sRecF = reinterpret_cast<float*>(sRec);
  const int tid = threadIdx.x;
  const int64_t o_base = int64_t(blockIdx.x) * NT;
  const int n_here = int(min<int64_t>(NT, n_obs - o_base));
  const bool act = tid < n_here;
  const int64_t o = act ? o_base + tid : o_base;
  {
    // the workgroup's Jacobian rows: two contiguous streams (main part, tail), scattered into the records
    const float4* __restrict__ src = reinterpret_cast<const float4*>(p.JpS) + 4 * o_base;
    for (int i = tid; i < 4 * n_here; i += NT) {  // piece i: observation i / 4, row (i / 2) & 1, entries 4 (i & 1) ...
      const float4 v = src[i];
      float* dst = sRecF + 2 * kA64Rec * (i >> 2) + 9 * ((i >> 1) & 1) + 4 * (i & 1);
      dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
    }
    for (int i = tid; i < 2 * n_here; i += NT) sRecF[2 * kA64Rec * (i >> 1) + 9 * (i & 1) + 8] = p.JpT[2 * o_base + i];
  }
  const int s = p.obs_lm[o];
  const float4* __restrict__ vh = reinterpret_cast<const float4*>(p.Vh);
  const float4 va = vh[2 * o], vb = vh[2 * o + 1];
  const int64_t o0 = p.lm_obs[s];
  const float4 w0 = vh[2 * o0], w1 = vh[2 * o0 + 1], w2 = vh[2 * o0 + 2];
  const double* __restrict__ lq = p.LQ + size_t(kA64Lq) * s;
  const double tau0 = lq[0], tau1 = lq[1], tau2 = lq[2], g10 = lq[3], g20 = lq[4], g21 = lq[5];
  const float* __restrict__ R = p.R0 + 6 * size_t(s);
  const float r0 = R[0], r1 = R[1], r2 = R[2], r3 = R[3], r4 = R[4], r5 = R[5];
  const int i = int(o - o0);
  // ---- the landmark's damping rotations (set_landmark_damping, landmark_block_base.ipp:165-210) in double ----
  double T[3][3] = {{double(r0), double(r1), double(r2)}, {0.0, double(r3), double(r4)}, {0.0, 0.0, double(r5)}};
  double D[3][3];
  const double sl = sqrt(lambda);
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) D[a][b] = (a == b) ? sl : 0.0;
  double gc[6], gs[6];
  {
    int idx = 0;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
#pragma unroll
      for (int m = 0; m <= n; ++m) {
        double c = 1.0, sn = 0.0;
        if (lambda != 0.0) make_givens<double>(T[n][n], D[n - m][n], c, sn);
        gc[idx] = c;
        gs[idx] = sn;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const double x = D[n - m][b], y = T[n][b];
          D[n - m][b] = c * x + sn * y;
          T[n][b] = -sn * x + c * y;
        }
        ++idx;
      }
    }
  }
  // ---- W' (3 x 2): the damped top rows of Q^T applied to the unit entries of the observation's two rows ----
  double W[3][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const double m0 = e == 0 ? 1.0 : 0.0, m1 = e == 0 ? 0.0 : 1.0;
    const double c0 = tau0 * (double(va.x) * m0 + double(vb.x) * m1);
    const double c1 = tau1 * (double(va.y) * m0 + double(vb.y) * m1 - c0 * g10);
    const double c2 = tau2 * (double(va.z) * m0 + double(vb.z) * m1 - c0 * g20 - c1 * g21);
    double tt[3] = {-(c0 * double(w0.x) + c1 * double(w0.y) + c2 * double(w0.z)),
                    -(c0 * double(w1.x) + c1 * double(w1.y) + c2 * double(w1.z)),
                    -(c0 * double(w2.x) + c1 * double(w2.y) + c2 * double(w2.z))};
    if (i == 0) {
      tt[0] += m0;
      tt[1] += m1;
    } else if (i == 1) {
      tt[2] += m0;
    }
    double d[3] = {0.0, 0.0, 0.0};
    int idx = 0;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
#pragma unroll
      for (int m = 0; m <= n; ++m) {
        const double x = d[n - m], y = tt[n];
        d[n - m] = gc[idx] * x + gs[idx] * y;
        tt[n] = -gs[idx] * x + gc[idx] * y;
        ++idx;
      }
    }
    W[0][e] = tt[0];
    W[1][e] = tt[1];
    W[2][e] = tt[2];
  }
  // ---- A: Cholesky of M = I - W'^T W' with the larger diagonal entry as pivot (kernels_cam.hpp) ----
  {
    const double m00 = 1.0 - (W[0][0] * W[0][0] + W[1][0] * W[1][0] + W[2][0] * W[2][0]);
    const double m01 = -(W[0][0] * W[0][1] + W[1][0] * W[1][1] + W[2][0] * W[2][1]);
    const double m11 = 1.0 - (W[0][1] * W[0][1] + W[1][1] * W[1][1] + W[2][1] * W[2][1]);
    double a00, a01, a10, a11;
    if (m00 >= m11) {
      a00 = sqrt(fmax(m00, 0.0));
      a01 = a00 > 0.0 ? m01 / a00 : 0.0;
      a10 = 0.0;
      a11 = sqrt(fmax(m11 - a01 * a01, 0.0));
    } else {
      a11 = sqrt(fmax(m11, 0.0));
      a10 = a11 > 0.0 ? m01 / a11 : 0.0;
      a01 = 0.0;
      a00 = sqrt(fmax(m00 - a10 * a10, 0.0));
    }
    if (act) {
      double2* dst = reinterpret_cast<double2*>(p.A + 4 * o);
      dst[0] = double2{a00, a01};
      dst[1] = double2{a10, a11};
    }
  }
  {
    double* r = sRec + kA64Rec * tid + kA64RecW;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      r[2 * n] = W[n][0];
      r[2 * n + 1] = W[n][1];
    }
    r[6] = 0.0;
  }
  __syncthreads();
  {
    double2* dst = reinterpret_cast<double2*>(p.rec + size_t(kA64Rec) * o_base);
    const double2* src = reinterpret_cast<const double2*>(sRec);
    for (int q = tid; q < n_here * (kA64Rec / 2); q += NT) dst[q] = src[q];
  }
}

// Off-diagonal blocks, one workgroup per block {c, d}, c < d, over its list of observation pairs (the double twin of
// k_ex_offdiag_mfma for the records of this file): - D_c [sum_pairs (W'_i Jp_i)^T (W'_j Jp_j)] D_d on
// v_mfma_f64_16x16x4_f64, four pairs per three instructions. The eight records of a quad of pairs are eight cache
// lines, fetched with one 16-byte load per lane (lane = record x piece) and staged in LDS; an operand of the
// instruction is two FMAs on two doubles (W') and two floats (Jp) of a staged record. The block is written where it
// is stored: as S_cd in row c and / or transposed in row d (half storage, kernels_pcg.hpp).
__global__ __launch_bounds__(256) void k_a64_offdiag(A64Params p, double* __restrict__ vals,
                                                     const int* __restrict__ upper_slot,
                                                     const int* __restrict__ mirror_slot,
                                                     const int64_t* __restrict__ pair_ptr,
                                                     const int* __restrict__ pair_oi, const int* __restrict__ pair_oj,
                                                     int n_upper) {
  using M = Mfma<double>;
  using Acc = typename M::acc;
  constexpr int U = 2;  // quads of pairs per step and wavefront (round-5 form: 4 -> 999 us at 116 registers, 2 -> 915 us at
                        // 72 registers = seven wavefronts per SIMD; round-4 form: 2 and 4 the same, 8 1.6 x slower)
  // Counters (profiles/r5_pmc_a64_offdiag.csv, round 5): per v_mfma_f64_16x16x4 the kernel issues 23 VALU, 4.4 LDS and
  // 3 scalar instructions; the LDS is its busiest unit (SQ_LDS_IDX_ACTIVE 59 % of the CU cycles, 9 % of that bank
  // conflicts), the VALU ~46 %, the matrix pipe 32 % (a v_mfma_f64_16x16x4 holds it for 64 cycles: 12.5 M of them are
  // exactly the 803 M busy cycles counted) - no unit is saturated, a wavefront is a chain of LDS round trip -> operand
  // arithmetic -> dependent matrix instruction. Four accumulator chains instead of two (160 instead of 128 registers,
  // one wavefront per SIMD less): 1160 -> 1287 us, reverted. The ISA of that form (scripts/pcgp_regs.sh writes
  // /tmp/solver.s) showed where the 23 VALU instructions came from: ~10 of operand arithmetic, the rest 64-bit index
  // clamps, the zero-selects of the staging in EVERY step, copies of the second accumulator between VGPRs and AGPRs,
  // and an exec-masked branch around every operand - the form below has one accumulator, 32-bit positions, selects
  // in a list's last step only, no predicate on the operands and two register sets instead of copies of load results:
  // 14.5 VALU instructions per matrix instruction measured (~11 in the loop body), 1152 -> 999 us; with two quads per
  // step instead of four 890-915 us - and the LDS 72 % busy: the bound then (profiles/r5_pmc_a64_offdiag_after.csv);
  // with the 2 x 2 matrix per pair (below) 681-686 us.
  __shared__ double tile[4][16][16];
  __shared__ __attribute__((aligned(16))) double stage[4][U][8][kA64Rec];
  __shared__ __attribute__((aligned(16))) double mbuf[4][U][4][4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int u = xcd_swizzled_camera(n_upper);
  if (u >= n_upper) return;
  const int i = lane & 15, kk = lane >> 4;
  // operand rows / columns 9 .. 15 of the 16 x 16 tile are never read back: their lanes build operands from the entries
  // of column 8 instead of zeros - no predicate, no branch around the operand arithmetic (round 5: the predicated form
  // compiled to an exec-masked block per matrix instruction - LDS reads, wait, arithmetic, instruction, twelve times in
  // a row with nothing hoisted across)
  const int i9 = min(i, 8);
  Acc acc = {0, 0, 0, 0};
  const int64_t q0 = pair_ptr[u], q1 = pair_ptr[u + 1];
  const int n = int(q1 - q0);  // pairs of this block; positions below are relative to q0 (32-bit index arithmetic)
  const int rec = lane >> 3, vec = lane & 7;
  const int* __restrict__ pair_side = (rec < 4 ? pair_oi : pair_oj) + q0;
  const int rsub = rec & 3;
  // Software pipeline (a wavefront's step is two DEPENDENT gathers - pair indices, then records that mostly miss L2 -
  // in front of 12 matrix-core instructions; without it the kernel ran at a third of the matrix rate with the matrix
  // pipe idle two thirds of the time, profiles/r4_pmc_mfma_float32.csv): the records of step s + 1 and the indices of
  // step s + 2 are in flight while step s is staged and multiplied. Clamped, not predicated: every load is issued.
  const int qw = wave * (4 * U);
  auto load_idx = [&](int q, int o[U]) {
#pragma unroll
    for (int uq = 0; uq < U; ++uq) o[uq] = pair_side[min(q + 4 * uq + rsub, n - 1)];
  };
  auto load_rec = [&](const int o[U], double2 v[U]) {
#pragma unroll
    for (int uq = 0; uq < U; ++uq) v[uq] = reinterpret_cast<const double2*>(p.rec + size_t(kA64Rec) * o[uq])[vec];
  };
  // Formulation (second step of round 5, after the counters showed the LDS as the bound at 4.4 instructions per matrix
  // instruction): sum_pairs (W'_i Jp_i)^T (W'_j Jp_j) = sum_pairs Jp_i^T M_ij Jp_j with the 2 x 2 matrix
  // M_ij = W'_i^T W'_j formed ONCE per pair (32 lanes, one value each, through LDS) - the inner dimension of a pair is
  // 2 instead of 3 (two matrix instructions per quad of pairs instead of three), the A operand is Jp_i^T M (two converted
  // floats, two doubles of M), the B operand a converted float of Jp_j with no arithmetic.
  static_assert(U == 2, "the M_ij of a step are formed by 16 U = 32 lanes");
  auto step = [&](int q, const double2 v[U]) {
    if (q + 4 * U > n) {  // (wave-uniform) the last step of the list: pairs beyond its end are staged as zeros
#pragma unroll
      for (int uq = 0; uq < U; ++uq) {
        const bool ok = q + 4 * uq + rsub < n;
        *reinterpret_cast<double2*>(&stage[wave][uq][rec][2 * vec]) = ok ? v[uq] : double2{0.0, 0.0};
      }
    } else {
#pragma unroll
      for (int uq = 0; uq < U; ++uq) *reinterpret_cast<double2*>(&stage[wave][uq][rec][2 * vec]) = v[uq];
    }
    wave_lds_fence();
    {
      // M[r][e] = sum_c W'_i[c][r] W'_j[c][e] of pair (uqm, pm): lane -> (quad, pair, r, e); lanes 32 .. 63 repeat
      const int l5 = lane & 31;
      const int uqm = l5 >> 4, pm = (l5 >> 2) & 3, rm = (l5 >> 1) & 1, em = l5 & 1;
      const double* wi = stage[wave][uqm][pm] + kA64RecW;
      const double* wj = stage[wave][uqm][4 + pm] + kA64RecW;
      const double mv = fma(wi[rm], wj[em], fma(wi[2 + rm], wj[2 + em], wi[4 + rm] * wj[4 + em]));
      if (lane < 32) mbuf[wave][uqm][pm][2 * rm + em] = mv;
    }
    wave_lds_fence();
#pragma unroll
    for (int uq = 0; uq < U; ++uq)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int g = 4 * m + kk;  // inner index 0..7 = (pair of the quad, row e of Jp_j)
        const int pp = g >> 1, e = g & 1;
        const float* fi = reinterpret_cast<const float*>(stage[wave][uq][pp]);
        const float* fj = reinterpret_cast<const float*>(stage[wave][uq][4 + pp]);
        const double* mm = mbuf[wave][uq][pp];
        const double av = fma(double(fi[i9]), mm[e], double(fi[9 + i9]) * mm[2 + e]);
        const double bv = double(fj[9 * e + i9]);
        acc = M::mma(av, bv, acc);
      }
    wave_lds_fence();  // the next step overwrites the staging buffers
  };
  // two register sets, the loop unrolled by two: no load result is ever copied (a copy waits for its load)
  int o_a[U], o_b[U];
  double2 v_a[U] = {}, v_b[U] = {};
  if (n > 0) {
    load_idx(qw, o_a);
    load_idx(qw + 16 * U, o_b);
    load_rec(o_a, v_a);
  }
  for (int q = qw; q < n; q += 32 * U) {
    load_rec(o_b, v_b);
    load_idx(q + 32 * U, o_a);
    step(q, v_a);
    if (q + 16 * U >= n) break;
    load_rec(o_a, v_a);
    load_idx(q + 48 * U, o_b);
    step(q + 16 * U, v_b);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[wave][M::row(lane, r)][i] = acc[r];
  __syncthreads();
  if (threadIdx.x < 81 && q1 > q0) {
    const int a = threadIdx.x / 9, b = threadIdx.x - 9 * a;
    const int ci = p.obs_cam[pair_oi[q0]], cj = p.obs_cam[pair_oj[q0]];
    const double t = ((tile[0][a][b] + tile[1][a][b]) + tile[2][a][b]) + tile[3][a][b];
    const double v = -t * double(p.pose_scaling[9 * ci + a]) * double(p.pose_scaling[9 * cj + b]);
    const int us = upper_slot[u], m = mirror_slot[u];
    if (us >= 0) vals[size_t(81) * us + threadIdx.x] = v;
    if (m >= 0) vals[size_t(81) * m + 9 * b + a] = v;
  }
}

// Diagonal blocks, one workgroup per camera (the double twin of k_cam_pass_mfma<S, 0>'s K part): gathers the float
// Jacobian rows (72 B) and the double factor A (32 B) of the camera's observations, Y = A Jp in double, K = sum Y^T Y
// on v_mfma_f64_16x16x4_f64 (two observations per instruction), vals[diag] = D K D (no pose damping: the product adds
// lambda x).
// GRAM = true: A = I, i.e. the blocks D (sum Jp^T Jp) D = D Hpp D of the JACOBI and power-series preconditioners, in
// double, into a dense [n_c][81] array (diag_slot == nullptr): a float Cholesky of the float-accumulated Hpp + lambda I
// met non-positive pivots on final-13682; sums and factorisation in double do not
// (k_invert_blocks<S, double>).
template <bool GRAM>
__global__ __launch_bounds__(256) void k_a64_diag(A64Params p, const int* __restrict__ diag_slot,
                                                  double* __restrict__ vals) {
  using M = Mfma<double>;
  using Acc = typename M::acc;
  constexpr int CH = kCamChunk, RW = 22;  // staged record: [Jp 18 | A 4]
  __shared__ __attribute__((aligned(16))) double stage[4][CH * RW];
  __shared__ double tile[4][16][16];
  const int c = xcd_swizzled_camera(p.n_cams);
  if (c >= p.n_cams) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t t0 = p.cam_obs_off[c], t1 = p.cam_obs_off[c + 1];
  Acc accK = {0, 0, 0, 0};
  const int i = lane & 15, kk = lane >> 4, i9 = min(i, 8);
  double* lds = stage[wave];
  int idxreg = t1 > t0 ? p.cam_obs[min<int64_t>(t0 + CH * wave + lane, t1 - 1)] : 0;
  for (int64_t base = t0 + CH * wave; base < t1; base += 4 * CH) {
    const int cnt = int(min<int64_t>(CH, t1 - base));
    const int idxnext = p.cam_obs[min<int64_t>(base + 4 * CH + lane, t1 - 1)];
    // an observation's rows: ONE aligned 64-byte line (four 16-byte pieces) + the two tail entries
    constexpr int NJ = CH * 4 / 64;
    static_assert(CH * 4 % 64 == 0 && CH <= 32, "whole passes of 16-byte pieces; one tail pair per lane");
    float4 jv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int q = j * 64 + lane;
      const int o = __shfl(idxreg, (q >> 2) & 31);
      jv[j] = reinterpret_cast<const float4*>(p.JpS)[int64_t(o) * 4 + (q & 3)];
    }
    const float2 jt = reinterpret_cast<const float2*>(p.JpT)[__shfl(idxreg, lane & 31)];
    double2 w = {0.0, 0.0};
    if (!GRAM) {
      const int o = __shfl(idxreg, (lane >> 1) & 31);
      w = *reinterpret_cast<const double2*>(p.A + int64_t(o) * 4 + 2 * (lane & 1));
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int q = j * 64 + lane;
      const int r = q >> 2, pc = q & 3;
      if (r < cnt) {
        double* d = lds + r * RW + 9 * (pc >> 1) + 4 * (pc & 1);
        d[0] = double(jv[j].x), d[1] = double(jv[j].y), d[2] = double(jv[j].z), d[3] = double(jv[j].w);
      }
    }
    if (lane < cnt) {
      lds[lane * RW + 8] = double(jt.x);
      lds[lane * RW + 17] = double(jt.y);
    }
    if (!GRAM) {
      const int r = lane >> 1, h = lane & 1;
      if (r < cnt) *reinterpret_cast<double2*>(lds + r * RW + 18 + 2 * h) = w;
    }
    wave_lds_fence();
    // (operands unpredicated, as in k_a64_offdiag: lanes of the tile's rows 9 .. 15 repeat column 8, records past the
    //  chunk's end are read - stale LDS - and replaced by zeros with a select)
    for (int s = 0; s < cnt; s += 4) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int so = s + 2 * h + (kk >> 1);
        const double* rec = lds + so * RW;
        const double* arow = rec + 18 + 2 * (kk & 1);
        double v = GRAM ? rec[9 * (kk & 1) + i9] : fma(arow[0], rec[i9], arow[1] * rec[9 + i9]);
        v = so < cnt ? v : 0.0;
        accK = M::mma(v, v, accK);
      }
    }
    wave_lds_fence();  // the next chunk overwrites the staging buffer
    idxreg = idxnext;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[wave][M::row(lane, r)][lane & 15] = accK[r];
  __syncthreads();
  if (tid < 81) {
    const int ii = tid / 9, jj = tid - 9 * ii;
    const double t = (tile[0][ii][jj] + tile[1][ii][jj]) + (tile[2][ii][jj] + tile[3][ii][jj]);
    vals[size_t(81) * (diag_slot ? diag_slot[c] : c) + tid] =
        t * double(p.pose_scaling[9 * c + ii]) * double(p.pose_scaling[9 * c + jj]);
  }
}

}  // namespace rba
