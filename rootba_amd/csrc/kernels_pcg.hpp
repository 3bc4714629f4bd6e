// kernels_pcg.hpp — PCG on the ASSEMBLED reduced camera matrix (block-CSR, kernels_sc.hpp) in
// TWO launches per iteration.
//
// The Ceres-style PCG (ConjugateGradientsSolver::solve, src/rootba/cg/conjugate_gradient.hpp:113-298)
// has two global reductions per iteration (rho = r.z before the direction update, p.q before the
// step), so two grid-wide synchronisation points are the minimum for the unmodified recurrence.
// Round 1 spent seven launches per iteration (44 us on venice-1778, 37 % of all kernel time); here
// each reduction boundary is one kernel boundary and everything else is fused around it:
//
//   k_pcgs_spmv<MODE 0>  [one wavefront per block-row item]
//        prologue (every wavefront, identical arithmetic => identical decisions):
//          Q-model termination test of the PREVIOUS iteration (sum of 64 partials), rho (sum of 64
//          partials), beta;
//        p_j = z_j + beta p_old_j evaluated on the fly for the row's column blocks (p is
//          ping-ponged: the owner of row c publishes p_new[c] into the other buffer);
//        q_c = sum_j S_cj p_j + lambda p_c; partial of p.q per item.
//   k_pcgs_update        [64 workgroups, 28 cameras per 252-thread tile]
//        p.q (fixed-order sum of the item partials), alpha; x += alpha p; r -= alpha q;
//        z = M^-1 r (block-diagonal preconditioner); partials of rho = r.z and Q = -x.(b + r).
//   every residual_reset_period-th iteration (conjugate_gradient.hpp:230-235):
//        k_pcgs_update (x only) -> k_pcgs_spmv<MODE 1> (S x) -> k_pcgs_update (r = b - S x, z, partials).
//
// SpMV data path: the 81-scalar blocks of a row are contiguous in HBM/MALL, so a wavefront copies
// up to 64 blocks (20.7 KB in float) with 16-byte fully coalesced loads into LDS and then every
// lane multiplies ITS block out of LDS (lane stride 81 words: odd, conflict-free). No integer
// division, no predicated 9-way accumulate, 21 independent 16-byte loads in flight per lane.
// Rows with more than 64*kSpmvChunksPerItem blocks are split into several items whose partial
// sums the update kernel adds in a fixed order; every reduction has a fixed order, so all ranks of a
// sharded run compute bit-identical iterates from the (all-reduced, identical) matrix.
//
// State hand-over between the two kernels never reads a field that the SAME kernel writes
// (CgState: `iter`, `need_test` are written by the update kernel only, `cur`, rho_hist[], q_hist[],
// `beta` by the SpMV kernel only); `done` is written by whoever detects termination - all
// workgroups detect it themselves from the same partial sums, a late workgroup that already sees
// `done` returns just as it would have on its own.
#pragma once

#include "kernels_sc.hpp"

namespace rba {

constexpr int kSpmvChunksPerItem = 4;  // 64-block chunks walked by one wavefront

struct SpmvItem {
  int row;    // camera (block row)
  int slot0;  // first block slot of the item
  int slot1;  // one past the last
  int first;  // 1: first item of its row (adds lambda p_c, publishes p_new[c])
};

// vector of 16 bytes of scalars
template <class S>
struct Vec16;
template <>
struct Vec16<float> {
  typedef float type __attribute__((ext_vector_type(4)));
  static constexpr int N = 4;
};
template <>
struct Vec16<double> {
  typedef double type __attribute__((ext_vector_type(2)));
  static constexpr int N = 2;
};

template <class S>
constexpr size_t spmv_lds_bytes() {
  return (size_t(64) * 81 + Vec16<S>::N) * sizeof(S);
}

// y[0..8] (all lanes) += S_row,chunk * v for the blocks [slot0, slot1) of one row; v is read by
// the callback (lane j gets the 9 entries of column block j)
template <class S, class XF>
__device__ __forceinline__ void spmv_item_accumulate(const int* __restrict__ cols, const S* __restrict__ vals,
                                                     int slot0, int slot1, S* lds, int lane, double acc[9],
                                                     XF&& load_x) {
  using V = typename Vec16<S>::type;
  constexpr int N = Vec16<S>::N;
  constexpr int PASS = 21;  // 16-byte loads in flight per lane
  for (int chunk = slot0; chunk < slot1; chunk += 64) {
    const int nb = min(64, slot1 - chunk);
    const bool act = lane < nb;
    const int col = act ? cols[chunk + lane] : 0;
    S xv[9];
    load_x(col, act, xv);
    // the chunk's scalars [g0, g0 + 81 nb) -> LDS at the same 16-byte phase
    const int64_t g0 = int64_t(81) * chunk;
    const int64_t base = g0 & ~int64_t(N - 1);
    const int off = int(g0 - base);
    const int nvec = (off + 81 * nb + N - 1) / N;
    const V* __restrict__ src = reinterpret_cast<const V*>(vals + base);
    V* dst = reinterpret_cast<V*>(lds);
    for (int v0 = 0; v0 < nvec; v0 += 64 * PASS) {
      V tmp[PASS];
#pragma unroll
      for (int u = 0; u < PASS; ++u) {
        const int i = v0 + u * 64 + lane;
        if (i < nvec) tmp[u] = src[i];
      }
#pragma unroll
      for (int u = 0; u < PASS; ++u) {
        const int i = v0 + u * 64 + lane;
        if (i < nvec) dst[i] = tmp[u];
      }
    }
    __syncthreads();
    if (act) {
      const S* blk = lds + off + 81 * lane;
#pragma unroll
      for (int a = 0; a < 9; ++a) {
        S t = S(0);
#pragma unroll
        for (int b = 0; b < 9; ++b) t += blk[9 * a + b] * xv[b];
        acc[a] += double(t);
      }
    }
    __syncthreads();  // the next chunk overwrites the staging buffer
  }
}

// MODE 0: direction update + product + p.q partial.   MODE 1: refresh product S x (+ lambda x).
template <class S, int MODE>
__global__ __launch_bounds__(64) void k_pcgs_spmv(const int* __restrict__ cols, const S* __restrict__ vals,
                                                  const SpmvItem* __restrict__ items, const S* __restrict__ z,
                                                  S* pbuf0, S* pbuf1, const S* __restrict__ xvec,
                                                  S* __restrict__ qpart, S lambda, CgState* st,
                                                  const double* __restrict__ part_rho,
                                                  const double* __restrict__ part_q,
                                                  double* __restrict__ part_pq, double q_tolerance, int min_it,
                                                  int max_it, int period, int* host_progress) {
  extern __shared__ __attribute__((aligned(16))) char smem_pcgs[];
  S* lds = reinterpret_cast<S*>(smem_pcgs);
  const int lane = threadIdx.x;
  if (st->done) return;
  const SpmvItem item = items[blockIdx.x];
  const int c = item.row;
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};

  if (MODE == 1) {
    if (st->cur % period != 0) return;
    spmv_item_accumulate<S>(cols, vals, item.slot0, item.slot1, lds, lane, acc,
                            [&](int col, bool act, S xv[9]) {
#pragma unroll
                              for (int t = 0; t < 9; ++t) xv[t] = act ? xvec[9 * col + t] : S(0);
                            });
    S mine = S(0);
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      const double tot = wave_sum(acc[a]);
      if (lane == a) mine = S(tot);
    }
    if (lane < 9) {
      if (item.first) mine += lambda * xvec[9 * c + lane];
      qpart[9 * blockIdx.x + lane] = mine;
    }
    return;
  }

  // ---- prologue: termination test of the previous iteration, rho, beta ------------
  const int it = st->iter;  // iterations completed
  const double rho = wave_sum(part_rho[lane]);
  const double q1 = wave_sum(part_q[lane]);
  const int need_test = st->need_test;
  int stop = 0, term = 0, res_it = it;
  if (need_test) {
    // Q-model test (conjugate_gradient.hpp:239-276); residual-based test is off (r_tolerance = -1)
    const double zeta = it * (q1 - st->q_hist[(it + 1) & 1]) / q1;
    if (zeta < q_tolerance && it >= min_it) {
      stop = 1;
      term = 1;
    } else if (it >= max_it) {
      stop = 1;
      term = 0;
    }
  }
  double beta = 0.0;
  if (!stop) {
    if (rho == 0.0 || isinf(rho)) {
      stop = 1;
      term = 2;  // "Numerical failure. rho / beta"
      res_it = it + 1;
    } else if (it > 0) {
      beta = rho / st->rho_hist[(it + 1) & 1];
      if (beta == 0.0 || isinf(beta)) {
        stop = 1;
        term = 2;
        res_it = it + 1;
      }
    }
  }
  if (blockIdx.x == 0 && lane == 0) {
    if (need_test) st->q_hist[it & 1] = q1;
    if (stop) {
      st->termination = term;
      st->result_iter = res_it;
      st->done = 1;
      if (host_progress) __hip_atomic_store(host_progress + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      st->rho_hist[it & 1] = rho;
      st->beta = beta;
      st->cur = it + 1;
      if (host_progress) __hip_atomic_store(host_progress, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (stop) return;

  // ---- q_c = sum_j S_cj (z_j + beta p_j) ---------------------------------------------
  const S* __restrict__ p_old = (it & 1) ? pbuf1 : pbuf0;
  S* __restrict__ p_new = (it & 1) ? pbuf0 : pbuf1;
  const S bs = S(beta);
  const bool first_it = it == 0;
  spmv_item_accumulate<S>(cols, vals, item.slot0, item.slot1, lds, lane, acc,
                          [&](int col, bool act, S xv[9]) {
#pragma unroll
                            for (int t = 0; t < 9; ++t) {
                              const S zz = act ? z[9 * col + t] : S(0);
                              const S pp = (act && !first_it) ? p_old[9 * col + t] : S(0);
                              xv[t] = first_it ? zz : zz + bs * pp;
                            }
                          });
  S mine = S(0);
#pragma unroll
  for (int a = 0; a < 9; ++a) {
    const double tot = wave_sum(acc[a]);
    if (lane == a) mine = S(tot);
  }
  double pq = 0.0;
  if (lane < 9) {
    const S zz = z[9 * c + lane];
    const S pc = first_it ? zz : zz + bs * p_old[9 * c + lane];
    if (item.first) {
      mine += lambda * pc;  // pose damping term of right_multiply
      p_new[9 * c + lane] = pc;
    }
    qpart[9 * blockIdx.x + lane] = mine;
    pq = double(pc) * double(mine);
  }
  pq = wave_sum(pq);
  if (lane == 0) part_pq[blockIdx.x] = pq;
}

// phase 0: after the direction product; phase 1: after the refresh product
template <class S>
__global__ __launch_bounds__(256) void k_pcgs_update(const S* __restrict__ inv, const S* __restrict__ bvec,
                                                     S* __restrict__ x, S* __restrict__ r, S* __restrict__ z,
                                                     const S* pbuf0, const S* pbuf1, const S* __restrict__ qpart,
                                                     const int* __restrict__ item_ptr, int n_items, int n_cams,
                                                     CgState* st, const double* __restrict__ part_pq,
                                                     double* __restrict__ part_rho, double* __restrict__ part_q,
                                                     int phase, int period, int* host_progress) {
  __shared__ double sm[4];
  __shared__ S rl[252];
  if (st->done) return;
  const int tid = threadIdx.x;
  const int cur = st->cur;
  const bool refresh = (cur % period) == 0;
  if (phase == 1 && !refresh) return;
  S a = S(0);
  if (phase == 0) {
    double acc = 0;
    for (int i = tid; i < n_items; i += 256) acc += part_pq[i];
    const double pq = pcg_block_sum(acc, sm);
    int stop = 0, term = 0;
    double alpha = 0;
    if (pq <= 0.0 || isinf(pq)) {
      stop = 1;  // "Matrix is indefinite, no more progress can be made." -> NO_CONVERGENCE
    } else {
      alpha = st->rho_hist[(cur + 1) & 1] / pq;
      if (isinf(alpha)) {
        stop = 1;
        term = 2;
      }
    }
    if (blockIdx.x == 0 && tid == 0) {
      st->pq = pq;
      st->alpha = alpha;
      if (stop) {
        st->termination = term;
        st->indefinite = term == 0 ? 1 : 0;
        st->result_iter = cur;
        st->done = 1;
        if (host_progress) __hip_atomic_store(host_progress + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (stop) return;
    a = S(alpha);
  }
  const S* __restrict__ p = (cur & 1) ? pbuf1 : pbuf0;
  double acc_rho = 0, acc_q = 0;
  const int n_tiles = (n_cams + 27) / 28;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int i = 252 * tile + tid;
    const bool act = tid < 252 && i < 9 * n_cams;
    const int c = act ? i / 9 : 0, row = act ? i - 9 * c : 0;
    S xn = S(0), rn = S(0);
    if (act) {
      S qv = S(0);
      for (int q = item_ptr[c]; q < item_ptr[c + 1]; ++q) qv += qpart[9 * q + row];
      if (phase == 0) {
        xn = x[i] + a * p[i];
        x[i] = xn;
        if (!refresh) {
          rn = r[i] - a * qv;
          r[i] = rn;
        }
      } else {
        xn = x[i];
        rn = bvec[i] - qv;  // qv = (S x + lambda x)_i
        r[i] = rn;
      }
    }
    if (phase == 0 && refresh) continue;  // uniform: the residual comes from the refresh product
    __syncthreads();
    if (tid < 252) rl[tid] = rn;
    __syncthreads();
    if (act) {
      const S* M = inv + 81 * c + 9 * row;
      const S* rc = rl + 9 * (tid / 9);
      S zc = S(0);
#pragma unroll
      for (int j = 0; j < 9; ++j) zc += M[j] * rc[j];
      z[i] = zc;
      acc_rho += double(rn) * double(zc);
      acc_q -= double(xn) * double(bvec[i] + rn);
    }
  }
  if (phase == 0 && refresh) return;
  const double rho_p = pcg_block_sum(acc_rho, sm);
  const double q_p = pcg_block_sum(acc_q, sm);
  if (tid == 0) {
    part_rho[blockIdx.x] = rho_p;
    part_q[blockIdx.x] = q_p;
    if (blockIdx.x == 0) {
      st->iter = cur;
      st->need_test = 1;
    }
  }
}

// r = b - (S x + lambda x) from the item partials of a refresh product (operator switch inside a solve)
template <class S>
__global__ __launch_bounds__(256) void k_pcgs_residual(const S* __restrict__ bvec, S* __restrict__ r,
                                                       const S* __restrict__ qpart,
                                                       const int* __restrict__ item_ptr, int n,
                                                       const CgState* st) {
  if (st->done) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = i / 9, row = i - 9 * c;
  S qv = S(0);
  for (int q = item_ptr[c]; q < item_ptr[c + 1]; ++q) qv += qpart[9 * q + row];
  r[i] = bvec[i] - qv;
}

}  // namespace rba
