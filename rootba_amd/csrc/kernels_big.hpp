// kernels_big.hpp — landmarks with MORE than 112 observations (k-class "big").
//
// Real BAL problems contain a few very long tracks. Here ONE WORKGROUP (256 threads) owns a landmark and
// reduces each block row across the workgroup; landmark-sized vectors live in a global scratch of 8 scalars
// per block row, so there is no limit on k. Same math and the same output buffers as the wave-tile kernels;
// correctness path first - these landmarks are rare.
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

// E0 * v (same result as k_e0)
template <class S>
__global__ __launch_bounds__(256) void k_e0_big(Params<S> p, int lm_begin, const S* __restrict__ v,
                                                S* __restrict__ y,
                                                const int* __restrict__ done_flag, S* __restrict__ e0_w) {
  __shared__ S sm[4];
  if (done_flag && *done_flag) return;
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
  if (e0_w) {  // deterministic form (k_e0_det_gather)
    if (tid < 3) e0_w[3 * size_t(s) + tid] = tid == 0 ? w[0] : (tid == 1 ? w[1] : w[2]);
    return;
  }
  for (int j = tid; j < ncols; j += 256) {
    const int i = j / 9, comp = j - 9 * i;
    const S* t = Td + kTd * i + comp;
    atomic_add(y + 9 * p.obs_cam[o0 + i] + comp, t[0] * w[0] + t[9] * w[1] + t[18] * w[2]);
  }
}


// ===========================================================================
// Long tracks in the implicit-Q configuration: NO dense block, NO landmark-sized LDS vectors, hence
// no limit on the number of observations of a landmark (the reference's dynamic block takes any k,
// landmark_block_dynamic.hpp:49-69). One workgroup per landmark, rows strided over the 256 threads;
// the per-row working set lives in a small global scratch (8 scalars per block row of the long
// tracks only) that stays in L2. Same records and per-landmark scalars as the tiled kernels
// (kernels_s1.hpp), so the column pass, stage 2, the back-substitution and the assembly of the reduced
// matrix treat these landmarks like any other.
// ===========================================================================
template <class S>
__device__ __forceinline__ void big_block_sum3(S& a, S& b, S& c, S* sm) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const S ta = wave_sum(a), tb = wave_sum(b), tc = wave_sum(c);
  __syncthreads();
  if (lane == 0) {
    sm[wave] = ta;
    sm[4 + wave] = tb;
    sm[8 + wave] = tc;
  }
  __syncthreads();
  a = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  b = (sm[4] + sm[5]) + (sm[6] + sm[7]);
  c = (sm[8] + sm[9]) + (sm[10] + sm[11]);
}

// pass Q for one long track (same outputs as s1_qr_tile). scratch: [rows][8] = jl[3], rs, vm[3], -
template <class S>
__global__ __launch_bounds__(256) void k_s1_qr_big(Params<S> p, int lm_begin, S* __restrict__ scratch,
                                                   const int64_t* __restrict__ scratch_off) {
  using V4 = typename std::conditional<sizeof(S) == 4, float4, double4>::type;
  __shared__ S sm[12];
  const int tid = threadIdx.x;
  const int s = lm_begin + blockIdx.x;
  const int nrows = 2 * p.lm_k[s];
  const int64_t row0 = 2 * p.lm_obs[s];
  S* W = scratch + 8 * scratch_off[blockIdx.x];
  // Jl column scaling (scale_Jl_cols)
  S s0 = S(0), s1 = S(0), s2 = S(0);
  for (int r = tid; r < nrows; r += 256) {
    const V4 v = reinterpret_cast<const V4*>(p.Vh)[row0 + r];
    s0 += v.x * v.x;
    s1 += v.y * v.y;
    s2 += v.z * v.z;
  }
  big_block_sum3(s0, s1, s2, sm);
  const S sc0 = S(1) / (p.eps + sqrt(s0)), sc1 = S(1) / (p.eps + sqrt(s1)), sc2 = S(1) / (p.eps + sqrt(s2));
  if (tid == 0) {
    p.jl_scale[3 * s + 0] = sc0;
    p.jl_scale[3 * s + 1] = sc1;
    p.jl_scale[3 * s + 2] = sc2;
  }
  for (int r = tid; r < nrows; r += 256) {
    const V4 v = reinterpret_cast<const V4*>(p.Vh)[row0 + r];
    S* w = W + 8 * size_t(r);
    w[0] = v.x * sc0;
    w[1] = v.y * sc1;
    w[2] = v.z * sc2;
    w[3] = v.w;
    w[4] = w[5] = w[6] = S(0);
    S* dst = p.JlS + 3 * (row0 + r);
    dst[0] = w[0];
    dst[1] = w[1];
    dst[2] = w[2];
    p.rS[row0 + r] = v.w;
  }
  __syncthreads();
  S tau[3], R[6] = {S(0), S(0), S(0), S(0), S(0), S(0)};
  for (int m = 0; m < 3; ++m) {
    const S c0 = W[8 * size_t(m) + m];
    S tail = S(0), z1 = S(0), z2 = S(0);
    for (int r = tid; r < nrows; r += 256)
      if (r > m) tail += W[8 * size_t(r) + m] * W[8 * size_t(r) + m];
    big_block_sum3(tail, z1, z2, sm);
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
    // reflector entries, and its dot products with the remaining Jl columns and the residual
    S d1 = S(0), d2 = S(0), d3 = S(0);
    for (int r = tid; r < nrows; r += 256) {
      S* w = W + 8 * size_t(r);
      const S vm = (r == m) ? S(1) : (r > m ? w[m] * inv : S(0));
      w[4 + m] = vm;
      if (m == 0) {
        d1 += vm * w[1];
        d2 += vm * w[2];
      } else if (m == 1) {
        d2 += vm * w[2];
      }
      d3 += vm * w[3];
    }
    big_block_sum3(d1, d2, d3, sm);
    d1 *= tau[m];
    d2 *= tau[m];
    d3 *= tau[m];
    for (int r = tid; r < nrows; r += 256) {
      S* w = W + 8 * size_t(r);
      const S vm = w[4 + m];
      if (m == 0) {
        w[1] -= d1 * vm;
        w[2] -= d2 * vm;
      } else if (m == 1) {
        w[2] -= d2 * vm;
      }
      w[3] -= d3 * vm;
      if (r == m) w[m] = beta;
      if (r > m) w[m] = S(0);
    }
    __syncthreads();
  }
  S g10 = S(0), g20 = S(0), g21 = S(0), d0 = S(0), d1 = S(0), d2 = S(0);
  for (int r = tid; r < nrows; r += 256) {
    const S* w = W + 8 * size_t(r);
    g10 += w[5] * w[4];
    g20 += w[6] * w[4];
    g21 += w[6] * w[5];
    if (r >= 3) {
      d0 += w[4] * w[3];
      d1 += w[5] * w[3];
      d2 += w[6] * w[3];
    }
    reinterpret_cast<V4*>(p.Vh)[row0 + r] = V4{w[4], w[5], w[6], w[3]};
  }
  big_block_sum3(g10, g20, g21, sm);
  big_block_sum3(d0, d1, d2, sm);
  if (tid == 0) {
    R[0] = W[0];
    R[1] = W[1];
    R[2] = W[2];
    R[3] = W[8 + 1];
    R[4] = W[8 + 2];
    R[5] = W[16 + 2];
    S* Ro = p.R0 + 6 * s;
#pragma unroll
    for (int i = 0; i < 6; ++i) Ro[i] = R[i];
    p.tauH[3 * s + 0] = tau[0];
    p.tauH[3 * s + 1] = tau[1];
    p.tauH[3 * s + 2] = tau[2];
    V4* lq = reinterpret_cast<V4*>(p.LQ + 12 * size_t(s));
    lq[0] = V4{tau[0], tau[1], tau[2], g10};
    lq[1] = V4{g20, g21, d0, d1};
    lq[2] = V4{d2, S(0), S(0), S(0)};
  }
}

// second pass of the back-substitution (k_bs_landmark) for one long track: one workgroup, strided sums
template <class S>
__global__ __launch_bounds__(256) void k_bs_landmark_big(Params<S> p, int lm_begin) {
  __shared__ S sm[12];
  const int tid = threadIdx.x;
  const int s = lm_begin + blockIdx.x;
  const int64_t ob = p.lm_obs[s], oe = p.lm_obs[s + 1];
  S r0 = S(0), r1 = S(0), r2 = S(0);
  for (int64_t o = ob + tid; o < oe; o += 256) {
    r0 += p.bsO[5 * o];
    r1 += p.bsO[5 * o + 1];
    r2 += p.bsO[5 * o + 2];
  }
  big_block_sum3(r0, r1, r2, sm);
  const S rhs[3] = {p.q1trd[3 * s] + r0, p.q1trd[3 * s + 1] + r1, p.q1trd[3 * s + 2] + r2};
  const S* Rd = p.Rd + 6 * s;
  S inc[3];
  inc[2] = rhs[2] / Rd[5];
  inc[1] = (rhs[1] - Rd[4] * inc[2]) / Rd[3];
  inc[0] = (rhs[0] - Rd[1] * inc[1] - Rd[2] * inc[2]) / Rd[0];
#pragma unroll
  for (int m = 0; m < 3; ++m) inc[m] = -inc[m];
  S acc = S(0), z1 = S(0), z2 = S(0);
  for (int64_t o = ob + tid; o < oe; o += 256) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const S* jl = p.JlS + 6 * o + 3 * r;
      const S v = p.bsO[5 * o + 3 + r] + jl[0] * inc[0] + jl[1] * inc[1] + jl[2] * inc[2];
      acc += v * (S(0.5) * v + p.rS[2 * o + r]);
    }
  }
  big_block_sum3(acc, z1, z2, sm);
  if (tid == 0) finish_landmark(p, s, inc, acc);
}

// H*x from the factors for one long track (same operator as hx_implicit_tile): u = Jp x per row in the
// scratch, the six reflections as strided passes with a workgroup reduction each, Z on the three top
// rows, then y_obs = Jp_obs^T u_obs scatter-added per observation.
template <class S>
__global__ __launch_bounds__(256) void k_hx_implicit_big(Params<S> p, int lm_begin, S* __restrict__ scratch,
                                                         const int64_t* __restrict__ scratch_off,
                                                         const S* __restrict__ x, S* __restrict__ y,
                                                         const S* __restrict__ dout,
                                                         const int* __restrict__ done_flag, S* __restrict__ hx_u) {
  using V4 = typename std::conditional<sizeof(S) == 4, float4, double4>::type;
  __shared__ S sm[12];
  if (done_flag && *done_flag) return;
  const int tid = threadIdx.x;
  const int s = lm_begin + blockIdx.x;
  const int nrows = 2 * p.lm_k[s];
  const int64_t o0 = p.lm_obs[s], row0 = 2 * o0;
  S* U = scratch + 8 * scratch_off[blockIdx.x];  // one scalar per row is used
  const V4* __restrict__ vh = reinterpret_cast<const V4*>(p.Vh);
  const S tau[3] = {p.tauH[3 * s], p.tauH[3 * s + 1], p.tauH[3 * s + 2]};
  S a = S(0), z1 = S(0), z2 = S(0);
  for (int r = tid; r < nrows; r += 256) {
    const int cam = p.obs_cam[o0 + (r >> 1)];
    S jrow[9];
    jp_row<S>(p.JpS, p.JpT, row0 + r, jrow);
    S u = S(0);
#pragma unroll
    for (int c = 0; c < 9; ++c) u += jrow[c] * x[9 * cam + c];
    U[r] = u;
    a += vh[row0 + r].x * u;
  }
  // W^T: reflectors 0, 1, 2; then Z on the top rows; then W: reflectors 2, 1, 0
  const int order[6] = {0, 1, 2, 2, 1, 0};
  for (int step = 0; step < 6; ++step) {
    const int m = order[step];
    big_block_sum3(a, z1, z2, sm);  // sum of v_m . u (also orders the scratch traffic of the passes)
    const S d = tau[m] * a;
    a = S(0);
    if (step == 2) {
      // apply reflector 2, then the 3x3 map Z to the top three entries, and start the sum for reflector 2
      for (int r = tid; r < nrows; r += 256) {
        const V4 v = vh[row0 + r];
        U[r] -= d * v.z;
      }
      __syncthreads();
      if (tid == 0) {
        const S u0 = U[0], u1 = U[1], u2 = U[2];
        const S* Z = p.Zd + 9 * s;
        U[0] = Z[0] * u0 + Z[1] * u1 + Z[2] * u2;
        U[1] = Z[3] * u0 + Z[4] * u1 + Z[5] * u2;
        U[2] = Z[6] * u0 + Z[7] * u1 + Z[8] * u2;
      }
      __syncthreads();
      for (int r = tid; r < nrows; r += 256) a += vh[row0 + r].z * U[r];
    } else {
      const int next = step < 5 ? order[step + 1] : -1;
      for (int r = tid; r < nrows; r += 256) {
        const V4 v = vh[row0 + r];
        const S vm = m == 0 ? v.x : (m == 1 ? v.y : v.z);
        const S u = U[r] - d * vm;
        U[r] = u;
        if (next >= 0) a += (next == 0 ? v.x : (next == 1 ? v.y : v.z)) * u;
      }
    }
  }
  __syncthreads();
  if (hx_u) {  // deterministic form (k_hx_det_gather, kernels.hpp)
    for (int r = tid; r < nrows; r += 256) hx_u[row0 + r] = U[r];
    return;
  }
  // y_obs = Jp_obs^T u_obs
  for (int j = tid; j < 9 * (nrows / 2); j += 256) {
    const int i = j / 9, c = j - 9 * i;
    // (entry c of the observation's two rows in the split storage: kernels.hpp, jp_row)
    const int64_t w0 = 2 * (o0 + i);
    const S j0 = c < 8 ? p.JpS[8 * w0 + c] : p.JpT[w0], j1 = c < 8 ? p.JpS[8 * w0 + 8 + c] : p.JpT[w0 + 1];
    const int yi = 9 * p.obs_cam[o0 + i] + c;
    const S w = j0 * U[2 * i] + j1 * U[2 * i + 1];
    atomic_add(y + yi, dout ? w * dout[yi] : w);
  }
}

}  // namespace rba
