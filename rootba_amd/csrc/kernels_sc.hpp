// kernels_sc.hpp — block-CSR reduced camera matrix: (1) the explicit Schur-complement backend
// (solver_type = SCHUR_COMPLEMENT), (2) further down, the explicitly assembled matrix of the
// SQUARE-ROOT solver for long PCG solves, and the SpMV both share.
//
// (1) Schur-complement backend.
// What the reference's LinearizorSC / LinearizationSC / LandmarkBlockSC compute
// (src/rootba/solver/linearizor_sc.cpp:70-211, src/rootba/sc/linearization_sc.hpp:55-359,
// src/rootba/sc/landmark_block.hpp:127-446), laid out for the GPU instead of one
// (2k x 16) Eigen matrix per landmark + a concurrent hash map of 9x9 blocks:
//
//   linearize   per OBSERVATION (one thread each): sqrt(w) Jp D_p (18), sqrt(w) Jl (6),
//               sqrt(w) r (2), obs-major SoA records; per LANDMARK (one thread each,
//               fixed order): M = Jl^T Jl (6 unique), v = Jl^T r, column scale of Jl.
//   prepare(l)  per landmark: H_ll^-1 = (S M S + lambda I)^-1, H_ll^-1 b_l;
//               per observation: W = Jp^T Jl (9x3), T = W H_ll^-1, and the gradient
//               record Jp^T (r - Jl H_ll^-1 b_l); gradient summed camera-major (fixed order).
//   assemble    one wavefront per 9x9 block of the reduced matrix: S(ci, cj) = - sum over the
//               landmarks seen by both cameras of T_i W_j^T  (+ Jp_i^T Jp_i on the diagonal),
//               gathered through a per-block list of observation pairs; the block-CSR
//               structure (all co-observing camera pairs) and the lists are fixed at
//               construction (dense n_c x n_c slot table on the host).
//   S x         block-CSR SpMV, one workgroup per block row, no atomics.
//   back-sub    one thread per landmark (fixed order): delta = -H_ll^-1 Jl^T (r + Jp x),
//               l_diff -= J_inc^T (J_inc / 2 + r), p_w += delta o scale.
// The 81-float blocks are contiguous ([slot][a][b]); every reduction has a fixed order,
// the backend uses no atomics at all.
#pragma once

#include "kernels.hpp"

namespace rba {

template <class S>
struct ScParams {
  int n_cams, n_lms;
  int64_t n_obs;
  // entry a of row r of observation o: entries 0..7 at JpS + 8 (2 o + r), entry 8 behind the main part
  __device__ __forceinline__ int64_t jp(int64_t o, int r, int a) const {
    const int64_t w = 2 * o + r;
    return a < 8 ? 8 * w + a : 16 * n_obs + w;
  }
  // topology (shared with the square-root path)
  const int* lm_k;
  const int64_t* lm_obs;
  const int* obs_cam;
  const int* obs_lm;
  const S* obs_xy;
  const int64_t* cam_obs_off;
  const int* cam_obs;
  // state
  S* cams;
  S* lms;
  const S* pose_scaling;
  // linearisation records
  S* JpS;   // [obs][18]  sqrt(w) Jp D_p, in the split storage of the square-root solver's rows (kernels.hpp, jp_row): jp()
  S* JlS;   // [obs][6]   sqrt(w) Jl (column scale applied on use)
  S* rS;    // [obs][2]
  S* M;     // [lm][6]    Jl^T Jl: 00 01 02 11 12 22
  S* v;     // [lm][3]    Jl^T r
  S* scale; // [lm][3]    1 / (eps + |Jl col|)
  S* Hinv;  // [lm][9]
  S* hb;    // [lm][3]    H_ll^-1 b_l
  S* W;     // [obs][27]  Jp^T Jl
  S* T;     // [obs][27]  W H_ll^-1
  S* bO;    // [obs][9]
  // reduced system
  const int* row_ptr;  // [n_cams + 1] first slot of each block row
  const int* cols;     // [nnz] column camera of each slot
  const int* diag_slot;  // [n_cams]
  S* vals;             // [nnz][81]
  S* b;                // [9 n_cams]
  S* blocks;           // [n_cams][81] diagonal blocks (preconditioner input)
  int* fail_flag;
  double* lm_ldiff;
  int robust_norm, valid_only;
  S huber, eps;
};

// SC backend, stage 1, pass A (camera-major, geometry re-evaluated per observation): squared column norms of the weighted pose Jacobian
// (add_Jp_diag2, landmark_block_base.ipp:493-518) and the non-finite check of
// linearize_landmark (ipp:123-146).
template <class S>
__global__ __launch_bounds__(256) void k_sc_jp_diag2(Params<S> p) {
  __shared__ double sm4[4];
  const int c = blockIdx.x;
  S cam[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) cam[i] = p.cams[10 * c + i];
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool fin = true;
  for (int64_t t = p.cam_obs_off[c] + threadIdx.x; t < p.cam_obs_off[c + 1]; t += 256) {
    const int o = p.cam_obs[t];
    const int l = p.obs_lm[o];
    S res[2], Jp[18], Jl[6];
    const bool valid = linearize_obs<S>(cam, p.lms[3 * l], p.lms[3 * l + 1], p.lms[3 * l + 2],
                                        p.obs_xy[2 * o], p.obs_xy[2 * o + 1], res, Jp, Jl);
    if (p.valid_only && !valid) continue;
    fin = fin && is_finite(res[0]) && is_finite(res[1]);
#pragma unroll
    for (int i = 0; i < 18; ++i) fin = fin && is_finite(Jp[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) fin = fin && is_finite(Jl[i]);
    S err, w;
    error_weight<S>(p.robust_norm, p.huber, res[0] * res[0] + res[1] * res[1], err, w);
#pragma unroll
    for (int a = 0; a < 9; ++a) acc[a] += double(w * (Jp[a] * Jp[a] + Jp[9 + a] * Jp[9 + a]));
  }
  if (!fin) atomicOr(p.fail_flag, 1);
#pragma unroll
  for (int a = 0; a < 9; ++a) {
    const double t = block_sum_256(acc[a], sm4);
    if (threadIdx.x == 0) p.jp_diag2[9 * c + a] = S(t);
  }
}

// ---- linearize ----------------------------------------------------------------
// linearize_landmark + scale_Jp_cols (sc/landmark_block.hpp:127-176, 200-213), one
// thread per observation
template <class S>
__global__ __launch_bounds__(256) void k_sc_linearize_obs(ScParams<S> p) {
  const int64_t o = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (o >= p.n_obs) return;
  const int c = p.obs_cam[o], l = p.obs_lm[o];
  S cam[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) cam[i] = p.cams[10 * c + i];
  S res[2], Jp[18], Jl[6];
  const bool valid = linearize_obs<S>(cam, p.lms[3 * l], p.lms[3 * l + 1], p.lms[3 * l + 2],
                                      p.obs_xy[2 * o], p.obs_xy[2 * o + 1], res, Jp, Jl);
  S sw = S(0);
  if (!p.valid_only || valid) {
    bool fin = is_finite(res[0]) && is_finite(res[1]);
#pragma unroll
    for (int i = 0; i < 18; ++i) fin = fin && is_finite(Jp[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) fin = fin && is_finite(Jl[i]);
    if (!fin) atomicOr(p.fail_flag, 1);
    S err, w;
    error_weight<S>(p.robust_norm, p.huber, res[0] * res[0] + res[1] * res[1], err, w);
    sw = sqrt(w);
  }
  // invalid projections keep zero rows (storage_.setZero + skipped assignment)
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int a = 0; a < 9; ++a)
      p.JpS[p.jp(o, r, a)] = sw == S(0) ? S(0) : sw * Jp[9 * r + a] * p.pose_scaling[9 * c + a];
#pragma unroll
  for (int i = 0; i < 6; ++i) p.JlS[6 * o + i] = sw == S(0) ? S(0) : sw * Jl[i];
  p.rS[2 * o] = sw == S(0) ? S(0) : sw * res[0];
  p.rS[2 * o + 1] = sw == S(0) ? S(0) : sw * res[1];
}

// Jl^T Jl, Jl^T r and scale_Jl_cols (landmark_block.hpp:188-198), one thread per landmark
template <class S>
__global__ __launch_bounds__(256) void k_sc_landmark_moments(ScParams<S> p) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= p.n_lms) return;
  S m[6] = {0, 0, 0, 0, 0, 0}, v[3] = {0, 0, 0};
  for (int64_t o = p.lm_obs[l]; o < p.lm_obs[l + 1]; ++o) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const S j0 = p.JlS[6 * o + 3 * r], j1 = p.JlS[6 * o + 3 * r + 1], j2 = p.JlS[6 * o + 3 * r + 2];
      const S rr = p.rS[2 * o + r];
      m[0] += j0 * j0;
      m[1] += j0 * j1;
      m[2] += j0 * j2;
      m[3] += j1 * j1;
      m[4] += j1 * j2;
      m[5] += j2 * j2;
      v[0] += j0 * rr;
      v[1] += j1 * rr;
      v[2] += j2 * rr;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) p.M[6 * l + i] = m[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) p.v[3 * l + i] = v[i];
  p.scale[3 * l + 0] = S(1) / (p.eps + sqrt(m[0]));
  p.scale[3 * l + 1] = S(1) / (p.eps + sqrt(m[3]));
  p.scale[3 * l + 2] = S(1) / (p.eps + sqrt(m[5]));
}

// ---- prepare(lambda) ------------------------------------------------------------
// H_ll^-1 and H_ll^-1 b_l (add_Hb, landmark_block.hpp:222-232), one thread per landmark
template <class S>
__global__ __launch_bounds__(256) void k_sc_landmark_inverse(ScParams<S> p, S lambda) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= p.n_lms) return;
  const S s0 = p.scale[3 * l], s1 = p.scale[3 * l + 1], s2 = p.scale[3 * l + 2];
  const S* m = p.M + 6 * l;
  const S a00 = s0 * s0 * m[0] + lambda, a01 = s0 * s1 * m[1], a02 = s0 * s2 * m[2];
  const S a11 = s1 * s1 * m[3] + lambda, a12 = s1 * s2 * m[4], a22 = s2 * s2 * m[5] + lambda;
  // general 3x3 inverse by cofactors (Eigen's Matrix3::inverse())
  const S c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
  const S det = a00 * c00 + a01 * c01 + a02 * c02;
  const S id = S(1) / det;
  S h[9];
  h[0] = c00 * id;
  h[1] = c01 * id;
  h[2] = c02 * id;
  h[3] = h[1];
  h[4] = (a00 * a22 - a02 * a02) * id;
  h[5] = (a01 * a02 - a00 * a12) * id;
  h[6] = h[2];
  h[7] = h[5];
  h[8] = (a00 * a11 - a01 * a01) * id;
#pragma unroll
  for (int i = 0; i < 9; ++i) p.Hinv[9 * l + i] = h[i];
  const S v0 = s0 * p.v[3 * l], v1 = s1 * p.v[3 * l + 1], v2 = s2 * p.v[3 * l + 2];
  p.hb[3 * l + 0] = h[0] * v0 + h[1] * v1 + h[2] * v2;
  p.hb[3 * l + 1] = h[3] * v0 + h[4] * v1 + h[5] * v2;
  p.hb[3 * l + 2] = h[6] * v0 + h[7] * v1 + h[8] * v2;
}

// W = Jp^T Jl, T = W H_ll^-1, gradient record Jp^T (r - Jl H_ll^-1 b_l); thread per observation
template <class S>
__global__ __launch_bounds__(256) void k_sc_obs_products(ScParams<S> p) {
  const int64_t o = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (o >= p.n_obs) return;
  const int l = p.obs_lm[o];
  S jp[18], jl[6], h[9];
#pragma unroll
  for (int i = 0; i < 18; ++i) jp[i] = p.JpS[p.jp(o, i / 9, i % 9)];
#pragma unroll
  for (int i = 0; i < 6; ++i) jl[i] = p.JlS[6 * o + i] * p.scale[3 * l + (i % 3)];
#pragma unroll
  for (int i = 0; i < 9; ++i) h[i] = p.Hinv[9 * l + i];
  const S hb0 = p.hb[3 * l], hb1 = p.hb[3 * l + 1], hb2 = p.hb[3 * l + 2];
  const S t0 = p.rS[2 * o] - (jl[0] * hb0 + jl[1] * hb1 + jl[2] * hb2);
  const S t1 = p.rS[2 * o + 1] - (jl[3] * hb0 + jl[4] * hb1 + jl[5] * hb2);
#pragma unroll
  for (int a = 0; a < 9; ++a) {
    const S w0 = jp[a] * jl[0] + jp[9 + a] * jl[3];
    const S w1 = jp[a] * jl[1] + jp[9 + a] * jl[4];
    const S w2 = jp[a] * jl[2] + jp[9 + a] * jl[5];
    p.W[27 * o + 3 * a + 0] = w0;
    p.W[27 * o + 3 * a + 1] = w1;
    p.W[27 * o + 3 * a + 2] = w2;
    p.T[27 * o + 3 * a + 0] = w0 * h[0] + w1 * h[3] + w2 * h[6];
    p.T[27 * o + 3 * a + 1] = w0 * h[1] + w1 * h[4] + w2 * h[7];
    p.T[27 * o + 3 * a + 2] = w0 * h[2] + w1 * h[5] + w2 * h[8];
    p.bO[9 * o + a] = jp[a] * t0 + jp[9 + a] * t1;
  }
}

// b[c] = sum over the camera's observations (fixed order), one workgroup per camera
template <class S>
__global__ __launch_bounds__(256) void k_sc_cam_gradient(ScParams<S> p) {
  __shared__ double sm4[4];
  const int c = blockIdx.x;
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t t = p.cam_obs_off[c] + threadIdx.x; t < p.cam_obs_off[c + 1]; t += 256) {
    const int64_t o = p.cam_obs[t];
#pragma unroll
    for (int a = 0; a < 9; ++a) acc[a] += double(p.bO[9 * o + a]);
  }
#pragma unroll
  for (int a = 0; a < 9; ++a) {
    const double t = block_sum_256(acc[a], sm4);
    if (threadIdx.x == 0) p.b[9 * c + a] = S(t);
  }
}

// add_Hb (landmark_block.hpp:234-262), block-major: one workgroup per 9x9 block (ci <= cj)
// of the reduced matrix sums the contributions -T_i W_j^T of all landmarks seen by both
// cameras (plus Jp_i^T Jp_i on the diagonal) from a per-block list of observation pairs
// built at construction, and also writes the transposed block (cj, ci). Fixed order, no
// atomics, no zero fill; lane e owns entry e of the block, lanes 0..16 also entry 64 + e.
// The four wavefronts take interleaved chunks of kScUnroll pairs (the lists of a camera's
// own block and of neighbouring cameras are thousands long: one wavefront per block leaves
// a latency-bound tail), walk them with wave-uniform (scalar) index loads so that the
// 108-byte gathers of several pairs are in flight together, and are summed in wave order.
constexpr int kScUnroll = 4;
// three consecutive scalars with ONE load instruction (global_load_dwordx3 / 3 x dwordx2):
// the gathers are bound by the number of vector memory instructions, not by bytes
template <class S>
struct __attribute__((packed, aligned(sizeof(S)))) Triple {
  S v[3];
};
template <class S>
__device__ __forceinline__ void load3(const S* __restrict__ src, S out[3]) {
  const Triple<S> t = *reinterpret_cast<const Triple<S>*>(src);
  out[0] = t.v[0];
  out[1] = t.v[1];
  out[2] = t.v[2];
}
template <class S>
__global__ __launch_bounds__(256) void k_sc_assemble(ScParams<S> p, const int* __restrict__ upper_slot,
                                                     const int* __restrict__ mirror_slot,
                                                     const int64_t* __restrict__ pair_ptr,
                                                     const int* __restrict__ pair_oi,
                                                     const int* __restrict__ pair_oj, int n_upper) {
  __shared__ S part[3][81];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int u = blockIdx.x;
  const int a0 = lane / 9, b0 = lane - 9 * a0;  // entry lane       (a0 <= 7)
  const int e1 = 64 + lane;                      // entry 64 + lane  (lanes 0..16)
  const bool has1 = e1 < 81;
  const int a1 = has1 ? e1 / 9 : 0, b1 = has1 ? e1 - 9 * a1 : 0;
  S acc0 = S(0), acc1 = S(0);
  const int64_t q0 = pair_ptr[u], q1 = pair_ptr[u + 1];
  for (int64_t q = q0 + wave * kScUnroll; q < q1; q += 4 * kScUnroll) {
    int oi[kScUnroll], oj[kScUnroll];
    S t0[kScUnroll][3], w0[kScUnroll][3], t1[kScUnroll][3], w1[kScUnroll][3];
#pragma unroll
    for (int r = 0; r < kScUnroll; ++r) {
      const bool ok = q + r < q1;
      oi[r] = ok ? pair_oi[q + r] : -1;
      oj[r] = ok ? pair_oj[q + r] : -1;
    }
#pragma unroll
    for (int r = 0; r < kScUnroll; ++r) {
      const bool ok = oi[r] >= 0;
      const S* __restrict__ T = p.T + 27 * int64_t(ok ? oi[r] : 0);
      const S* __restrict__ W = p.W + 27 * int64_t(ok ? oj[r] : 0);
      load3(T + 3 * a0, t0[r]);
      load3(W + 3 * b0, w0[r]);
      load3(T + 3 * a1, t1[r]);
      load3(W + 3 * b1, w1[r]);
      if (!ok) t0[r][0] = t0[r][1] = t0[r][2] = S(0);
      if (!ok || !has1) t1[r][0] = t1[r][1] = t1[r][2] = S(0);
    }
#pragma unroll
    for (int r = 0; r < kScUnroll; ++r) {
      acc0 -= t0[r][0] * w0[r][0] + t0[r][1] * w0[r][1] + t0[r][2] * w0[r][2];
      acc1 -= t1[r][0] * w1[r][0] + t1[r][1] * w1[r][1] + t1[r][2] * w1[r][2];
      if (oi[r] >= 0 && oi[r] == oj[r]) {  // wave-uniform: diagonal blocks only
        const S* __restrict__ J = p.JpS;
        const int64_t oo = oi[r];
        acc0 += J[p.jp(oo, 0, a0)] * J[p.jp(oo, 0, b0)] + J[p.jp(oo, 1, a0)] * J[p.jp(oo, 1, b0)];
        if (has1) acc1 += J[p.jp(oo, 0, a1)] * J[p.jp(oo, 0, b1)] + J[p.jp(oo, 1, a1)] * J[p.jp(oo, 1, b1)];
      }
    }
  }
  if (wave > 0) {
    part[wave - 1][lane] = acc0;
    if (has1) part[wave - 1][e1] = acc1;
  }
  __syncthreads();
  if (wave > 0) return;
  acc0 = ((acc0 + part[0][lane]) + part[1][lane]) + part[2][lane];
  if (has1) acc1 = ((acc1 + part[0][e1]) + part[1][e1]) + part[2][e1];
  S* out = p.vals + size_t(81) * upper_slot[u];
  out[lane] = acc0;
  if (has1) out[e1] = acc1;
  const int m = mirror_slot[u];
  if (m >= 0) {
    S* outT = p.vals + size_t(81) * m;
    outT[9 * b0 + a0] = acc0;
    if (has1) outT[9 * b1 + a1] = acc1;
  }
}

// float version on the matrix cores (see k_ex_offdiag_mfma below for the reasoning): the block
// is  sum_pairs ( [i == j] Jp_i^T Jp_i - T_i W_j^T ),  two GEMMs over the pair list with one
// 4-byte gather per lane and operand per v_mfma_f32_16x16x4_f32.
__global__ __launch_bounds__(256) void k_sc_assemble_mfma(ScParams<float> p, const int* __restrict__ upper_slot,
                                                          const int* __restrict__ mirror_slot,
                                                          const int64_t* __restrict__ pair_ptr,
                                                          const int* __restrict__ pair_oi,
                                                          const int* __restrict__ pair_oj) {
  __shared__ float tile[4][16][16];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int u = blockIdx.x;
  const int i = lane & 15, kk = lane >> 4;
  const bool diagonal = mirror_slot[u] < 0;  // every pair of a diagonal block is (o, o)
  f32x4 accN = {0.f, 0.f, 0.f, 0.f}, accP = {0.f, 0.f, 0.f, 0.f};
  const int64_t q0 = pair_ptr[u], q1 = pair_ptr[u + 1];
  for (int64_t q = q0 + wave * 4; q < q1; q += 16) {
    int oi[4], oj[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = q + r < q1;
      oi[r] = ok ? pair_oi[q + r] : -1;
      oj[r] = ok ? pair_oj[q + r] : -1;
    }
    float av[3], bv[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int g = 4 * m + kk;  // inner index 0..11 = (pair, factor column)
      const int pp = g / 3, c = g - 3 * pp;
      const int o_i = pp == 0 ? oi[0] : pp == 1 ? oi[1] : pp == 2 ? oi[2] : oi[3];
      const int o_j = pp == 0 ? oj[0] : pp == 1 ? oj[1] : pp == 2 ? oj[2] : oj[3];
      const bool ok = i < 9 && o_i >= 0;
      av[m] = ok ? p.T[27 * int64_t(o_i) + 3 * i + c] : 0.f;
      bv[m] = ok ? p.W[27 * int64_t(o_j) + 3 * i + c] : 0.f;
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) accN = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[m], accN, 0, 0, 0);
    if (diagonal) {
      // Jp^T Jp of the four observations: inner index = (pair, Jacobian row), two instructions
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int g = 4 * m + kk;  // 0..7
        const int pp = g >> 1, row = g & 1;
        const int o = pp == 0 ? oi[0] : pp == 1 ? oi[1] : pp == 2 ? oi[2] : oi[3];
        const float v = (i < 9 && o >= 0) ? p.JpS[p.jp(o, row, i)] : 0.f;
        accP = __builtin_amdgcn_mfma_f32_16x16x4f32(v, v, accP, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[wave][kk * 4 + r][i] = accP[r] - accN[r];
  __syncthreads();
  if (threadIdx.x < 81) {
    const int a = threadIdx.x / 9, b = threadIdx.x - 9 * a;
    const float v = ((tile[0][a][b] + tile[1][a][b]) + tile[2][a][b]) + tile[3][a][b];
    p.vals[size_t(81) * upper_slot[u] + threadIdx.x] = v;
    const int m = mirror_slot[u];
    if (m >= 0) p.vals[size_t(81) * m + 9 * b + a] = v;
  }
}

// pose damping on the diagonal blocks (linearization_sc.hpp:323-327) and a copy of the
// diagonal blocks for the SCHUR_JACOBI preconditioner (linearizor_sc.cpp:134-137)
template <class S>
__global__ __launch_bounds__(256) void k_sc_damp_and_extract_diag(ScParams<S> p, S lambda) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 81 * p.n_cams) return;
  const int c = i / 81, e = i - 81 * c;
  S* blk = p.vals + size_t(81) * p.diag_slot[c];
  S v = blk[e];
  if (e % 10 == 0) {
    v += lambda;
    blk[e] = v;
  }
  p.blocks[i] = v;
}

// blocks = Hpp + lambda I from the (already scaled) Gram blocks: the blocks of the power-series preconditioner
// on the explicit system (LinearizationSC::get_jacobi, linearizor_sc.cpp:163-170)
template <class S>
__global__ __launch_bounds__(256) void k_sc_jacobi_blocks(const S* __restrict__ gram, S lambda, S* __restrict__ blocks,
                                                          int n_cams) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 81 * n_cams) return;
  blocks[i] = gram[i] + ((i % 81) % 10 == 0 ? lambda : S(0));
}

// ---- S x ------------------------------------------------------------------------
// BlockSparseMatrix::right_multiply: one workgroup per block row, threads over the
// row's contiguous 81 nnz_i entries; the 9 row sums are selected by predication
template <class S>
__global__ __launch_bounds__(256) void k_sc_spmv(ScParams<S> p, const S* __restrict__ x, S* __restrict__ y,
                                                 const int* __restrict__ done_flag) {
  __shared__ double part[4][9];
  if (done_flag && *done_flag) return;
  const int c = blockIdx.x;
  const int s0 = p.row_ptr[c], s1 = p.row_ptr[c + 1];
  const S* __restrict__ vals = p.vals + size_t(81) * s0;
  const int total = 81 * (s1 - s0);
  S acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int e = threadIdx.x; e < total; e += 256) {
    const int t = e / 81, rem = e - 81 * t;
    const int a = rem / 9, bb = rem - 9 * a;
    const S prod = vals[e] * x[9 * p.cols[s0 + t] + bb];
#pragma unroll
    for (int q = 0; q < 9; ++q) acc[q] += (q == a) ? prod : S(0);
  }
  // nine row sums: wave reductions, then one barrier and a fixed-order sum over the four waves
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const double t = wave_sum(double(acc[q]));
    if (lane == 0) part[wave][q] = t;
  }
  __syncthreads();
  if (threadIdx.x < 9)
    y[9 * c + threadIdx.x] =
        S((part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
}

// ---- back-substitution (landmark_block.hpp:409-446), one thread per landmark -----
template <class S>
__global__ __launch_bounds__(256) void k_sc_back_substitute(ScParams<S> p, const S* __restrict__ x) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= p.n_lms) return;
  const S s0 = p.scale[3 * l], s1 = p.scale[3 * l + 1], s2 = p.scale[3 * l + 2];
  const int64_t ob = p.lm_obs[l], oe = p.lm_obs[l + 1];
  S tmp[3] = {0, 0, 0};
  for (int64_t o = ob; o < oe; ++o) {
    const int c = p.obs_cam[o];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      S jp_inc = S(0);
#pragma unroll
      for (int a = 0; a < 9; ++a) jp_inc += p.JpS[p.jp(o, r, a)] * x[9 * c + a];
      const S t = p.rS[2 * o + r] + jp_inc;
      tmp[0] += p.JlS[6 * o + 3 * r] * s0 * t;
      tmp[1] += p.JlS[6 * o + 3 * r + 1] * s1 * t;
      tmp[2] += p.JlS[6 * o + 3 * r + 2] * s2 * t;
    }
  }
  const S* h = p.Hinv + 9 * l;
  const S d0 = -(h[0] * tmp[0] + h[1] * tmp[1] + h[2] * tmp[2]);
  const S d1 = -(h[3] * tmp[0] + h[4] * tmp[1] + h[5] * tmp[2]);
  const S d2 = -(h[6] * tmp[0] + h[7] * tmp[1] + h[8] * tmp[2]);
  S acc = S(0);
  for (int64_t o = ob; o < oe; ++o) {
    const int c = p.obs_cam[o];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      S j_inc = S(0);
#pragma unroll
      for (int a = 0; a < 9; ++a) j_inc += p.JpS[p.jp(o, r, a)] * x[9 * c + a];
      j_inc += p.JlS[6 * o + 3 * r] * s0 * d0 + p.JlS[6 * o + 3 * r + 1] * s1 * d1 + p.JlS[6 * o + 3 * r + 2] * s2 * d2;
      acc += j_inc * (S(0.5) * j_inc + p.rS[2 * o + r]);
    }
  }
  p.lm_ldiff[l] = -double(acc);
  if (!is_finite(acc)) atomicOr(p.fail_flag, 2);
  p.lms[3 * l + 0] += d0 * s0;
  p.lms[3 * l + 1] += d1 * s1;
  p.lms[3 * l + 2] += d2 * s2;
}

// ===========================================================================
// Explicit reduced camera matrix of the SQUARE-ROOT solver:
//   S = sum_l A_l^T A_l,  A_l = (Q2^T Jp)_l  (the 2k x 9k blocks H*x streams),
// in the same block-CSR layout as above, for long PCG solves (solver.hip: pcg()).
// With Q = [Q1 Q2] orthogonal and Jp block diagonal per observation,
//   A_l^T A_l = Jp^T Jp - (Q1^T Jp)^T (Q1^T Jp),
// so an OFF-DIAGONAL block (ci != cj) is exactly  - topd_i^T topd_j : a rank-3 product
// of the damped top rows that stage 2 produced with orthogonal transformations only
// (no H_ll^-1, nothing to cancel) — gathered block-major like the SC assembly.
// The DIAGONAL blocks are the ones stage 2 computes for the SCHUR_JACOBI preconditioner
// (sum over the camera's observations of Jp^T Jp - top^T top plus the damping rows;
// both sums have thousands of terms, the rounding error of the difference is of the
// same order, eps sqrt(n), as that of accumulating Gram blocks).
// ===========================================================================
// Strictly upper blocks on the matrix cores (v_mfma_*_16x16x4 of either precision, Mfma<S>): the block is the GEMM
//  - [T_1 T_2 ...] [W_1 W_2 ...]^T   (9 x 3P)(3P x 9) over the P pairs of its list, fixed order, mirrored write.
template <class S>
__global__ __launch_bounds__(256) void k_ex_offdiag_mfma(const S* __restrict__ topd, S* __restrict__ vals,
                                                         const int* __restrict__ upper_slot,
                                                         const int* __restrict__ mirror_slot,
                                                         const int64_t* __restrict__ pair_ptr,
                                                         const int* __restrict__ pair_oi,
                                                         const int* __restrict__ pair_oj, int n_upper) {
  using M = Mfma<S>;
  using V4 = typename M::V4;
  using Acc = typename M::acc;
  __shared__ S tile[4][16][16];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-contiguous block order (workgroup b runs on XCD b % 8): consecutive upper blocks share their row camera, so
  // the records an XCD gathers are re-used out of its own L2 instead of being fetched once per XCD
  const int u = xcd_swizzled_camera(n_upper);
  if (u >= n_upper) return;
  const int i = lane & 15, kk = lane >> 4;
  const int i9 = min(i, 8);  // (rows / columns 9 .. 15 of the tile repeat column 8 and are never read back: no predicate)
  Acc acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  const int64_t q0 = pair_ptr[u], q1 = pair_ptr[u + 1];
  const int n = int(q1 - q0);  // pairs of this block; positions below are relative to q0
  // 16 pairs per wave and step. A record (one observation's damped top rows) is ONE 128-byte cache line in float: the
  // eight records of a quad of pairs are fetched with one four-scalar load per lane (lane = record x piece), staged in LDS and
  // read back in the operand layout of the matrix-core instruction. (Round 2 gathered 4 bytes per lane straight into
  // the operand registers: six 36-of-64-lane gather instructions per quad kept the kernel on the texture-address
  // path - 0.95 ms on venice whatever the record size or the block order.)
  // Round 5, as k_a64_offdiag (kernels_a64.hpp): the records of step s + 1 and the pair indices of step s + 2 are in
  // flight while step s is multiplied (two register sets, the loop unrolled by two), zero-selects of the staging in
  // a list's last step only.
  constexpr int U = sizeof(S) == 8 ? 2 : 4;
  __shared__ __attribute__((aligned(32))) S stage[4][U][8][kTd];
  const int rec = lane >> 3, vec = lane & 7;
  const int* __restrict__ pair_side = (rec < 4 ? pair_oi : pair_oj) + q0;
  const int rsub = rec & 3;
  const int qw = wave * (4 * U);
  // (clamped, not predicated: the U index loads go out together, then the U record loads - a load inside a
  //  conditional is a basic block of its own that waits for its operand and for everything issued before it)
  auto load_idx = [&](int q, int o[U]) {
#pragma unroll
    for (int uq = 0; uq < U; ++uq) o[uq] = pair_side[min(q + 4 * uq + rsub, n - 1)];
  };
  auto load_rec = [&](const int o[U], V4 v[U]) {
#pragma unroll
    for (int uq = 0; uq < U; ++uq) v[uq] = reinterpret_cast<const V4*>(topd + kTd * int64_t(o[uq]))[vec];
  };
  auto step = [&](int q, const V4 v[U]) {
    if (q + 4 * U > n) {  // (wave-uniform) the last step of the list
#pragma unroll
      for (int uq = 0; uq < U; ++uq) {
        const bool ok = q + 4 * uq + rsub < n;
        *reinterpret_cast<V4*>(&stage[wave][uq][rec][4 * vec]) = ok ? v[uq] : V4{0, 0, 0, 0};
      }
    } else {
#pragma unroll
      for (int uq = 0; uq < U; ++uq) *reinterpret_cast<V4*>(&stage[wave][uq][rec][4 * vec]) = v[uq];
    }
    wave_lds_fence();
#pragma unroll
    for (int uq = 0; uq < U; ++uq)
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int g = 4 * m + kk;  // inner index 0..11 = (pair of the quad, factor row)
        const int pp = g / 3, c = g - 3 * pp;
        const S av = stage[wave][uq][pp][9 * c + i9];
        const S bv = stage[wave][uq][4 + pp][9 * c + i9];
        if (sizeof(S) == 4 && (uq & 1))
          acc2 = M::mma(av, bv, acc2);  // (float: two chains; a double instruction holds the pipe for its whole latency)
        else
          acc = M::mma(av, bv, acc);
      }
    wave_lds_fence();  // the next step overwrites the staging buffer
  };
  int o_a[U], o_b[U];
  V4 v_a[U] = {}, v_b[U] = {};
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
  for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[wave][M::row(lane, r)][i] = acc[r];
  __syncthreads();
  if (threadIdx.x < 81) {
    const int a = threadIdx.x / 9, b = threadIdx.x - 9 * a;
    const S v = -(((tile[0][a][b] + tile[1][a][b]) + tile[2][a][b]) + tile[3][a][b]);
    // (half storage, kernels_pcg.hpp: the block lives in the row of its owner - as S_cd in row c OR as S_dc in row d;
    //  -1 = not stored there. Full storage of the explicit-SC backend: both.)
    const int us = upper_slot[u], m = mirror_slot[u];
    if (us >= 0) vals[size_t(81) * us + threadIdx.x] = v;
    if (m >= 0) vals[size_t(81) * m + 9 * b + a] = v;
  }
}

// diagonal blocks: stage 2 already holds them (+ lambda I) as the SCHUR_JACOBI
// preconditioner input, summed camera-major on the matrix cores and all-reduced
template <class S>
__global__ __launch_bounds__(256) void k_ex_set_diag(const S* __restrict__ blocks, const int* __restrict__ diag_slot,
                                                     S* __restrict__ vals, S lambda, int n_cams) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 81 * n_cams) return;
  const int c = t / 81, e = t - 81 * c;
  vals[size_t(81) * diag_slot[c] + e] = blocks[t] - (e % 10 == 0 ? lambda : S(0));
}

// the same from a buffer that holds the diagonal blocks themselves (JACOBI / power series)
template <class S>
__global__ __launch_bounds__(256) void k_ex_copy_diag(const S* __restrict__ sdiag, const int* __restrict__ diag_slot,
                                                      S* __restrict__ vals, int n_cams) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 81 * n_cams) return;
  vals[size_t(81) * diag_slot[t / 81] + t % 81] = sdiag[t];
}

}  // namespace rba
