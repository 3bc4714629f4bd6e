// kernels_big.hpp — landmarks with MORE than 112 observations (k-class "big").
//
// Real BAL problems contain a few very long tracks. Their blocks are large
// (2k x 9k: 0.9 MB at k = 113, 73 MB at k = 1000), so the one-wavefront-per-
// landmark kernels with register-resident column chunks do not apply; here ONE
// WORKGROUP (256 threads) owns a landmark, walks the 9k columns in passes of 28
// cameras (252 lanes), keeps the landmark-sized vectors (x, y, Q^T r, the three
// reflectors) in LDS and reduces each block row across the workgroup.
// Same math and the same output buffers as kernels.hpp; correctness path first —
// these landmarks are rare.  Limit: k <= kBigMaxK (LDS: 2 x 8k scalars).
#pragma once

#include "kernels.hpp"

namespace rba {

constexpr int kBigMaxK = 2048;

// sum over the 256-thread workgroup, result in every thread (sm: >= 4 doubles)
template <class S>
__device__ __forceinline__ S big_block_sum(S v, S* sm) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const S t = wave_sum(v);
  __syncthreads();
  if (lane == 0) sm[wave] = t;
  __syncthreads();
  return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// ---------------------------------------------------------------------------
// stage 1, pass B  (same outputs as k_linearize_qr)
// LDS: V[2k][4] = (Jl | res) -> (R-part / Q^T r), W[2k][4] = reflectors v0 v1 v2
// ---------------------------------------------------------------------------
template <class S>
__global__ __launch_bounds__(256) void k_linearize_qr_big(Params<S> p, int lm_begin) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ S sm[4];
  const int tid = threadIdx.x;
  const int s = lm_begin + blockIdx.x;
  const int k = p.lm_k[s];
  const int64_t o0 = p.lm_obs[s];
  const int nrows = 2 * k, ncols = 9 * k;
  S* V = reinterpret_cast<S*>(smem_raw);
  S* W = V + 4 * size_t(nrows);
  const S pwx = p.lms[3 * s], pwy = p.lms[3 * s + 1], pwz = p.lms[3 * s + 2];

  // geometry: one thread per observation; weighted Jp parked in JpS (unscaled)
  for (int i = tid; i < k; i += 256) {
    const int64_t o = o0 + i;
    const int cam = p.obs_cam[o];
    S res[2], Jp[18], Jl[6];
    const bool valid = linearize_obs<S>(p.cams + 10 * cam, pwx, pwy, pwz, p.obs_xy[2 * o],
                                        p.obs_xy[2 * o + 1], res, Jp, Jl);
    S sw = S(0);
    if (!p.valid_only || valid) {
      S err, w;
      error_weight<S>(p.robust_norm, p.huber, res[0] * res[0] + res[1] * res[1], err, w);
      sw = sqrt(w);
    }
    for (int c = 0; c < 18; ++c) p.JpS[o * 18 + c] = sw * Jp[c];
    for (int r = 0; r < 2; ++r) {
      V[4 * (2 * i + r) + 0] = sw * Jl[3 * r + 0];
      V[4 * (2 * i + r) + 1] = sw * Jl[3 * r + 1];
      V[4 * (2 * i + r) + 2] = sw * Jl[3 * r + 2];
      V[4 * (2 * i + r) + 3] = sw * res[r];
    }
  }
  __syncthreads();
  // Jl column scaling
  for (int c = 0; c < 3; ++c) {
    S ss = S(0);
    for (int r = tid; r < nrows; r += 256) ss += V[4 * r + c] * V[4 * r + c];
    ss = big_block_sum(ss, sm);
    const S sc = S(1) / (p.eps + sqrt(ss));
    for (int r = tid; r < nrows; r += 256) V[4 * r + c] *= sc;
    if (tid == 0) p.jl_scale[3 * s + c] = sc;
    __syncthreads();
  }
  // raw (weighted, column-scaled) Jl rows and residual for the back-substitution's l_diff
  for (int r = tid; r < nrows; r += 256) {
    S* dst = p.JlS + 3 * (2 * o0 + r);
    dst[0] = V[4 * r + 0];
    dst[1] = V[4 * r + 1];
    dst[2] = V[4 * r + 2];
    p.rS[2 * o0 + r] = V[4 * r + 3];
  }
  // Householder QR of Jl, applied to the remaining columns and to the residual
  S tau[3];
  for (int m = 0; m < 3; ++m) {
    const S c0 = V[4 * m + m];
    S tail = S(0);
    for (int r = tid; r < nrows; r += 256)
      if (r > m) tail += V[4 * r + m] * V[4 * r + m];
    tail = big_block_sum(tail, sm);
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
    for (int r = tid; r < nrows; r += 256)
      W[4 * r + m] = (r == m) ? S(1) : (r > m ? V[4 * r + m] * inv : S(0));
    __syncthreads();
    for (int c2 = m + 1; c2 < 4; ++c2) {  // columns m+1..2 of Jl, then the residual (3)
      S d = S(0);
      for (int r = tid; r < nrows; r += 256) d += W[4 * r + m] * V[4 * r + c2];
      d = tau[m] * big_block_sum(d, sm);
      for (int r = tid; r < nrows; r += 256) V[4 * r + c2] -= d * W[4 * r + m];
    }
    __syncthreads();
    for (int r = tid; r < nrows; r += 256) {
      if (r == m) V[4 * r + m] = beta;
      if (r > m) V[4 * r + m] = S(0);
    }
    __syncthreads();
  }
  S g10 = S(0), g20 = S(0), g21 = S(0);
  for (int r = tid; r < nrows; r += 256) {
    g10 += W[4 * r + 1] * W[4 * r + 0];
    g20 += W[4 * r + 2] * W[4 * r + 0];
    g21 += W[4 * r + 2] * W[4 * r + 1];
  }
  g10 = big_block_sum(g10, sm);
  g20 = big_block_sum(g20, sm);
  g21 = big_block_sum(g21, sm);
  if (tid == 0) {
    S* R = p.R0 + 6 * s;
    R[0] = V[0];
    R[1] = V[1];
    R[2] = V[2];
    R[3] = V[4 + 1];
    R[4] = V[4 + 2];
    R[5] = V[8 + 2];
  }
  for (int r = tid; r < nrows; r += 256) {
    S* vh = p.Vh + 4 * (2 * o0 + r);
    vh[0] = W[4 * r + 0];
    vh[1] = W[4 * r + 1];
    vh[2] = W[4 * r + 2];
    vh[3] = V[4 * r + 3];
  }
  if (tid == 0) {
    p.tauH[3 * s + 0] = tau[0];
    p.tauH[3 * s + 1] = tau[1];
    p.tauH[3 * s + 2] = tau[2];
  }

  // column passes of 28 cameras
  S* Ablk = p.A + p.lm_blk[s];
  S* T0 = p.top0 + 27 * o0;
  const int cam28 = tid / 9, comp = tid - 9 * cam28;
  for (int pass = 0; pass * 28 < k; ++pass) {
    const int i = pass * 28 + cam28;
    const bool act = tid < 252 && i < k;
    if (!act) continue;
    const int j = 9 * i + comp;
    const int cam = p.obs_cam[o0 + i];
    const S d = p.pose_scaling[9 * cam + comp];
    const S m0 = p.JpS[(o0 + i) * 18 + comp] * d;
    const S m1 = p.JpS[(o0 + i) * 18 + 9 + comp] * d;
    const S* wa = W + 4 * (2 * i);
    const S* wb = W + 4 * (2 * i + 1);
    const S c0 = tau[0] * (wa[0] * m0 + wb[0] * m1);
    const S c1 = tau[1] * (wa[1] * m0 + wb[1] * m1 - c0 * g10);
    const S c2 = tau[2] * (wa[2] * m0 + wb[2] * m1 - c0 * g20 - c1 * g21);
    S bm = S(0);
    for (int r = 0; r < nrows; ++r) {
      S val = -(c0 * W[4 * r] + c1 * W[4 * r + 1] + c2 * W[4 * r + 2]);
      if (r == 2 * i) val += m0;
      if (r == 2 * i + 1) val += m1;
      if (r < 3) {
        T0[27 * i + 9 * r + comp] = val;
      } else {
        Ablk[size_t(r - 3) * ncols + j] = val;
        bm += val * V[4 * r + 3];
      }
    }
    Ablk[size_t(nrows - 3) * ncols + j] = S(0);
    Ablk[size_t(nrows - 2) * ncols + j] = S(0);
    Ablk[size_t(nrows - 1) * ncols + j] = S(0);
    p.JpS[(o0 + i) * 18 + comp] = m0;
    p.JpS[(o0 + i) * 18 + 9 + comp] = m1;
    p.bmO[(o0 + i) * 9 + comp] = bm;
  }
}


// ---------------------------------------------------------------------------
// H*x (same result as k_hx): x and y of the landmark live in LDS
// ---------------------------------------------------------------------------
template <class S>
__global__ __launch_bounds__(256) void k_hx_big(Params<S> p, int lm_begin, const S* __restrict__ x,
                                                S* __restrict__ y,
                                                const int* __restrict__ done_flag) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ S sm[4];
  if (done_flag && *done_flag) return;
  y = scatter_replica(p, y);
  const int tid = threadIdx.x;
  const int s = lm_begin + blockIdx.x;
  const int k = p.lm_k[s];
  const int64_t o0 = p.lm_obs[s];
  const int nrows = 2 * k, ncols = 9 * k;
  const S* __restrict__ Ablk = p.A + p.lm_blk[s];
  S* xs = reinterpret_cast<S*>(smem_raw);
  S* ys = xs + ncols;
  for (int j = tid; j < ncols; j += 256) {
    const int i = j / 9;
    xs[j] = x[9 * p.obs_cam[o0 + i] + (j - 9 * i)];
    ys[j] = S(0);
  }
  __syncthreads();
  for (int r = 0; r < nrows; ++r) {
    const S* row = Ablk + size_t(r) * ncols;
    S d = S(0);
    for (int j = tid; j < ncols; j += 256) d += row[j] * xs[j];
    const S t = big_block_sum(d, sm);
    for (int j = tid; j < ncols; j += 256) ys[j] += row[j] * t;  // same thread, same j
  }
  for (int j = tid; j < ncols; j += 256) {
    const int i = j / 9;
    atomic_add(y + 9 * p.obs_cam[o0 + i] + (j - 9 * i), ys[j]);
  }
}

// E0 * v (same result as k_e0)
template <class S>
__global__ __launch_bounds__(256) void k_e0_big(Params<S> p, int lm_begin, const S* __restrict__ v,
                                                S* __restrict__ y,
                                                const int* __restrict__ done_flag) {
  __shared__ S sm[4];
  if (done_flag && *done_flag) return;
  y = scatter_replica(p, y);
  const int tid = threadIdx.x;
  const int s = lm_begin + blockIdx.x;
  const int k = p.lm_k[s];
  const int64_t o0 = p.lm_obs[s];
  const int ncols = 9 * k;
  const S* __restrict__ Td = p.topd + kTd * o0;
  S w[3] = {S(0), S(0), S(0)};
  for (int j = tid; j < ncols; j += 256) {
    const int i = j / 9, comp = j - 9 * i;
    const S xv = v[9 * p.obs_cam[o0 + i] + comp];
#pragma unroll
    for (int m = 0; m < 3; ++m) w[m] += Td[kTd * i + 9 * m + comp] * xv;
  }
#pragma unroll
  for (int m = 0; m < 3; ++m) w[m] = big_block_sum(w[m], sm);
  for (int j = tid; j < ncols; j += 256) {
    const int i = j / 9, comp = j - 9 * i;
    const S* t = Td + kTd * i + comp;
    atomic_add(y + 9 * p.obs_cam[o0 + i] + comp, t[0] * w[0] + t[9] * w[1] + t[18] * w[2]);
  }
}


}  // namespace rba
