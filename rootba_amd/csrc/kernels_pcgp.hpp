// kernels_pcgp.hpp — the PCG on the assembled reduced camera matrix as ONE persistent kernel with the matrix RESIDENT IN
// THE REGISTER FILES (round 5).
//
// Two launches per iteration (kernels_pcg.hpp) cost 10.1 + 6.3 + 0.9 us on venice-1778 for a 36 MB matrix that never
// leaves the last-level cache: the iteration is bound by two kernel boundaries and by streaming S through the fabric
// again and again, not by arithmetic (2 x 55 K blocks x 81 FMAs). The chip's register files hold 128 MB. So:
//
//   * every workgroup (512 work-items = 8 wavefronts = two per SIMD at <= 256 VGPRs, ONE workgroup per CU) owns a run of
//     consecutive block rows in FULL storage, one 9x9 double block per lane (162 VGPRs), loaded once per solve from the
//     half-storage matrix in HBM (the mirror blocks are transposed on the way in). 256 CUs x 512 lanes = 131 K blocks
//     (venice-1778: 108.8 K, 111.4 K with the rows padded to whole quads); a matrix that does not fit keeps the
//     two-launch path (Solver::build_pcgp_structure decides);
//   * full storage means q_c = sum_j S_cj p_j is complete inside the workgroup that owns row c: x, r, z = M^-1 r live in
//     the registers of the row's nine work-items for the whole solve, and the ONLY data that crosses workgroups per
//     iteration are the 9 n_c entries of z (each workgroup needs the rows its blocks multiply: its "staged columns",
//     ~100 for a banded matrix) and the workgroups' partial sums of the two dot products;
//   * there is NO grid barrier. Every exchanged word travels as an 8-byte {value, tag} granule written by one
//     write-through store (sc1) and read by L1-bypassing loads until its tag is this iteration's (the data is the flag;
//     scripts/microbench/grid_barrier.hip: an all-gather of 256 partial sums 3.2 us against 7.1 us for the cheapest
//     counter barrier pair and 14-17 us with release / acquire fences). Overwriting is safe without double buffering:
//     the two all-gathers of an iteration guard each other - nobody publishes exchange k + 1 before it has read EVERY
//     workgroup's granule of exchange k, which those workgroups wrote after reading exchange k - 1;
//   * every sum has a fixed order (quad sums on the DPP network, the row's quads in ascending order, the waves of a
//     workgroup, the workgroups' partial sums lane-strided then the DPP tree): bitwise reproducible and identical on all
//     ranks of a sharded run, like the two-launch path;
//   * every spin is bounded: a workgroup that waits ~1 s raises an abort word, everybody leaves, the host repeats the
//     solve on the two-launch path and stops using this kernel for the handle.
//
// The recurrence is ConjugateGradientsSolver::solve (src/rootba/cg/conjugate_gradient.hpp:113-298) as restated in
// kernels_pcg.hpp (k_pcgs_spmv<0> prologue = decisions, k_pcgs_update = step), including the residual refresh every
// `period` iterations (:230-235) and the Q-model stopping rule (:263-276).
#pragma once

#include "kernels_pcg.hpp"

namespace rba {

using pg_u32 = unsigned int;
using pg_u64 = unsigned long long;

constexpr int kPgThreads = 512;               // 8 wavefronts: two per SIMD, one workgroup per CU
constexpr int kPgMaxRows = kPgThreads / 9;    // nine row work-items per camera
constexpr int kPgQuads = kPgThreads / 4;
constexpr int kPgMaxGroups = 256;             // workgroups whose partial sums one wavefront gathers (4 per lane)
constexpr pg_u32 kPgSpinLimit = 1u << 20;     // sweeps of ~1 us before a workgroup gives up

struct PgWorkgroup {
  int row0, nrows;  // cameras row0 .. row0 + nrows - 1
  int ncols;        // distinct columns its blocks multiply (staged in LDS by work-items 0 .. ncols - 1)
  int pad;
};

template <class S>
struct PgParams {
  const PgWorkgroup* wg;           // [G]
  const int* lane_src;             // [G][512]  2 * slot + transposed of the lane's block in the half-storage matrix; -1: padding
  const unsigned short* lane_col;  // [G][512]  staged column (index into the workgroup's list) of the lane's block
  const int* stage_col;            // [G][512]  camera of staged column t; -1 beyond ncols
  const int* row_info;             // [n_c][3]  first quad of the row in its workgroup, quads, staged index of its own column
  const double* vals;              // half storage [nnz][81]
  const S* inv;                    // M^-1 [n_c][81]
  const S* b;
  S* x;                            // in: iterate after `iter` iterations; out: the solution
  const S* r_in;                   // residual (not read when the operator is switched: recomputed)
  const S* p_in;                   // direction of the last completed iteration
  pg_u64* zg;                      // [9 n_c W] granules of z
  pg_u64* xg;                      // [9 n_c W] granules of x (refresh product)
  pg_u64* part_rq;                 // [G][4]    partial sums of rho and Q
  pg_u64* part_pq;                 // [G][2]    partial sums of p.q
  CgState* st;
  int* host_progress;              // pinned: [1] done, [4] aborted
  pg_u32 tag_base;
  int G;
  int switch_operator;             // the solve ran matrix-free so far: r = b - (S + lambda I) x first (like the refresh)
  double q_tolerance;
  int min_it, max_it, period;
};

#define PG_AGENT __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ pg_u64 pg_ld(const pg_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, PG_AGENT); }
__device__ __forceinline__ void pg_st(pg_u64* p, pg_u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, PG_AGENT); }

// a scalar as tagged granules: float = one, double = two (high / low word)
__device__ __forceinline__ void pg_store(pg_u64* g, size_t i, float v, pg_u32 tag) {
  pg_st(g + i, (pg_u64(tag) << 32) | pg_u64(__float_as_uint(v)));
}
__device__ __forceinline__ void pg_store(pg_u64* g, size_t i, double v, pg_u32 tag) {
  const pg_u64 bits = pg_u64(__double_as_longlong(v));
  pg_st(g + 2 * i, (pg_u64(tag) << 32) | (bits >> 32));
  pg_st(g + 2 * i + 1, (pg_u64(tag) << 32) | (bits & 0xffffffffull));
}
__device__ __forceinline__ void pg_load(const pg_u64* g, size_t i, pg_u32 tag, bool& ok, float& v) {
  const pg_u64 x = pg_ld(g + i);
  ok &= pg_u32(x >> 32) == tag;
  v = __uint_as_float(pg_u32(x));
}
__device__ __forceinline__ void pg_load(const pg_u64* g, size_t i, pg_u32 tag, bool& ok, double& v) {
  const pg_u64 h = pg_ld(g + 2 * i), l = pg_ld(g + 2 * i + 1);
  ok &= pg_u32(h >> 32) == tag && pg_u32(l >> 32) == tag;
  v = __longlong_as_double((long long)((h << 32) | (l & 0xffffffffull)));
}
template <class S>
constexpr int pg_words() {
  return int(sizeof(S) / 4);
}

template <class S>
constexpr size_t pgp_lds_bytes() {
  return size_t(2) * kPgThreads * 9 * sizeof(S)  // pst, opx
         + size_t(kPgQuads) * 9 * sizeof(double)  // red
         + size_t(kPgThreads) * sizeof(S)         // rl
         + size_t(kPgThreads) * 9 * sizeof(S)     // minv
         + 16 * sizeof(double)                    // smw
         + 4 * sizeof(double)                     // bc
         + 16;                                    // flags
}

// sum over the four lanes of a quad, every lane receives it (fixed order: (l0 + l1) + (l2 + l3) up to commutation)
__device__ __forceinline__ double pg_quad_sum(double v) {
  v += dpp_mov0<0xb1>(v);  // quad_perm:[1,0,3,2]
  v += dpp_mov0<0x4e>(v);  // quad_perm:[2,3,0,1]
  return v;
}

template <class S>
__global__ __launch_bounds__(kPgThreads) void k_pcgp(PgParams<S> P) {
  extern __shared__ __attribute__((aligned(16))) char smem_pg[];
  S* pst = reinterpret_cast<S*>(smem_pg);                       // [512][9] direction p of the staged columns (kept across iterations)
  S* opx = pst + kPgThreads * 9;                                // [512][9] operand of the refresh product (x)
  double* red = reinterpret_cast<double*>(opx + kPgThreads * 9);  // [128][9] quad sums of the block products
  S* rl = reinterpret_cast<S*>(red + kPgQuads * 9);             // [512]   residual of the row work-items (z = M^-1 r)
  S* minv = rl + kPgThreads;                                    // [512][9] row `ra` of M^-1 of the row work-items
  double* smw = reinterpret_cast<double*>(minv + kPgThreads * 9);  // [8][2] wave sums
  double* bc = smw + 16;                                        // [4]     broadcast scalars
  int* sflag = reinterpret_cast<int*>(bc + 4);                  // [0] abort

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = blockIdx.x;
  const PgWorkgroup W = P.wg[g];
  const int src = P.lane_src[size_t(g) * kPgThreads + tid];
  const bool act = src >= 0;
  const int scol = P.lane_col[size_t(g) * kPgThreads + tid];
  const int mycol = P.stage_col[size_t(g) * kPgThreads + tid];
  const bool stager = mycol >= 0;
  const bool wave_stages = wave * 64 < W.ncols;  // (the staged columns are work-items 0 .. ncols - 1)
  const bool gatherer = wave == kPgThreads / 64 - 1;  // the last wavefront gathers the workgroups' partial sums
  const bool rowt = tid < 9 * W.nrows;
  const int rr = rowt ? tid / 9 : 0, ra = rowt ? tid - 9 * rr : 0;
  const int c = W.row0 + rr;
  CgState* st = P.st;
  if (st->done) return;  // (uniform over the grid: nobody writes the state before the end)
  if (tid == 0) sflag[0] = 0;

  // ---- the lane's block: 81 doubles, for the whole solve -----------------------------------------------------------
  double blk[81];
  {
    const double* v = P.vals + size_t(81) * size_t(act ? (src >> 1) : 0);
    const bool tr = (src & 1) != 0;
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
      for (int bb = 0; bb < 9; ++bb) {
        const double t = v[tr ? 9 * bb + a : 9 * a + bb];
        blk[9 * a + bb] = act ? t : 0.0;
      }
  }
  // ---- state -----------------------------------------------------------------------------------------------------
  int it = st->iter, need_test = st->need_test;
  double rho_prev = st->rho_hist[(it + 1) & 1], q_prev = st->q_hist[(it + 1) & 1];
  const S lambda = S(st->lambda);
  int q0 = 0, nq = 0, self = 0;
  S x_i = S(0), r_i = S(0), b_i = S(0);
  if (rowt) {
    q0 = P.row_info[3 * c];
    nq = P.row_info[3 * c + 1];
    self = P.row_info[3 * c + 2];
    x_i = P.x[9 * c + ra];
    b_i = P.b[9 * c + ra];
    if (!P.switch_operator) r_i = P.r_in[9 * c + ra];
#pragma unroll
    for (int j = 0; j < 9; ++j) minv[9 * tid + j] = P.inv[81 * c + 9 * ra + j];
  }
  if (stager) {
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      pst[9 * tid + a] = it > 0 ? P.p_in[9 * mycol + a] : S(0);
      if (P.switch_operator) opx[9 * tid + a] = P.x[9 * mycol + a];
    }
  }

  // q_c[ra] = sum_j S_cj v_j for the row work-items, v = the staged operand (LDS); ends behind a workgroup barrier
  auto product = [&](const S* opnd) -> double {
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (act) {
      // (column by column: nine accumulators and ONE operand entry live beside the 162 registers of the block)
#pragma unroll
      for (int bb = 0; bb < 9; ++bb) {
        const double pv = double(opnd[9 * scol + bb]);
#pragma unroll
        for (int a = 0; a < 9; ++a) acc[a] += blk[9 * a + bb] * pv;
      }
    }
#pragma unroll
    for (int a = 0; a < 9; ++a) acc[a] = pg_quad_sum(acc[a]);
    if ((lane & 3) == 0) {
#pragma unroll
      for (int a = 0; a < 9; ++a) red[9 * (tid >> 2) + a] = acc[a];
    }
    __syncthreads();
    double q = 0.0;
    if (rowt)
      for (int k = 0; k < nq; ++k) q += red[9 * (q0 + k) + ra];
    return q;
  };
  // sums over the workgroup of two per-work-item values; valid in work-item 0 (behind a workgroup barrier)
  auto wg_sum2 = [&](double v0, double v1, double& s0, double& s1) {
    const double t0 = wave_sum(v0), t1 = wave_sum(v1);
    if (lane == 0) {
      smw[2 * wave] = t0;
      smw[2 * wave + 1] = t1;
    }
    __syncthreads();
    s0 = ((smw[0] + smw[2]) + (smw[4] + smw[6])) + ((smw[8] + smw[10]) + (smw[12] + smw[14]));
    s1 = ((smw[1] + smw[3]) + (smw[5] + smw[7])) + ((smw[9] + smw[11]) + (smw[13] + smw[15]));
  };
  // z = M^-1 r, partial sums of rho = r.z and Q = -x.(b + r); published for the iteration with tag `tagn`
  auto close_residual = [&](pg_u32 tagn) {
    rl[tid] = r_i;
    __syncthreads();
    double acc_rho = 0.0, acc_q = 0.0;
    if (rowt) {
      const S* rc = rl + 9 * rr;
      S zc = S(0);
#pragma unroll
      for (int j = 0; j < 9; ++j) zc += minv[9 * tid + j] * rc[j];
      pg_store(P.zg, size_t(9) * c + ra, zc, tagn);
      acc_rho = double(r_i) * double(zc);
      acc_q = -double(x_i) * double(b_i + r_i);
    }
    double s0, s1;
    wg_sum2(acc_rho, acc_q, s0, s1);
    if (tid == 0) {
      pg_store(P.part_rq, size_t(2) * g, s0, tagn);
      pg_store(P.part_rq, size_t(2) * g + 1, s1, tagn);
    }
  };
  // what a wavefront does when its sweep did not find this exchange's tags: wait a little, give up after ~1 s
  auto spin_failed = [&](pg_u32& spins) -> bool {
    __builtin_amdgcn_s_sleep(1);
    ++spins;
    if ((spins & 1023u) == 0 && __hip_atomic_load(P.host_progress + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return true;
    return spins > kPgSpinLimit;
  };
  auto raise_abort = [&]() {
    if (lane == 0) {
      __hip_atomic_store(P.host_progress + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(sflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  // the solution and the final state (every workgroup takes the same decisions from the same sums). `stepped`: the
  // iteration it + 1 was started (direction and product done) when the step length came out unusable.
  auto finish = [&](int termination, int result_iter, int indefinite, bool stepped, double rho, double q1, double beta,
                    double pq, double alpha) {
    if (rowt) P.x[9 * c + ra] = x_i;
    if (g == 0 && tid == 0) {
      if (need_test) st->q_hist[it & 1] = q1;
      if (stepped) {
        st->rho_hist[it & 1] = rho;
        st->beta = beta;
        st->cur = it + 1;
        st->pq = pq;
        st->alpha = alpha;
      }
      st->iter = it;
      st->need_test = need_test;
      st->termination = termination;
      st->indefinite = indefinite;
      st->result_iter = result_iter;
      st->done = 1;
      __hip_atomic_store(P.host_progress + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  };

  __syncthreads();  // pst / opx / sflag
  if (P.switch_operator) {
    // operator switch inside a running solve: the residual is recomputed with the operator used from here on,
    // r = b - (S + lambda I) x, exactly like the periodic refresh (Solver::pcg_fused)
    const double q = product(opx);
    if (rowt) {
      S qs = S(q);
      qs += lambda * x_i;
      r_i = b_i - qs;
    }
  }
  close_residual(P.tag_base + pg_u32(it + 1));

  for (;;) {
    const int cur = it + 1;
    const pg_u32 tag = P.tag_base + pg_u32(cur);
    // ---- exchange 1: z of the staged columns, partial sums of rho and Q ---------------------------------------------
    S zv[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) zv[a] = S(0);
    if (wave_stages || gatherer) {
      pg_u32 spins = 0;
      double a_rho = 0.0, a_q = 0.0;
      for (;;) {
        bool ok = true;
        if (stager) {
#pragma unroll
          for (int a = 0; a < 9; ++a) pg_load(P.zg, size_t(9) * mycol + a, tag, ok, zv[a]);
        }
        if (gatherer) {
          a_rho = 0.0;
          a_q = 0.0;
#pragma unroll
          for (int k = 0; k < kPgMaxGroups / 64; ++k) {
            const int gg = lane + 64 * k;
            if (gg < P.G) {
              double v0, v1;
              pg_load(P.part_rq, size_t(2) * gg, tag, ok, v0);
              pg_load(P.part_rq, size_t(2) * gg + 1, tag, ok, v1);
              a_rho += v0;
              a_q += v1;
            }
          }
        }
        if (__ballot(!ok) == 0) break;
        if (spin_failed(spins)) {
          raise_abort();
          break;
        }
      }
      if (gatherer) {
        const double t0 = wave_sum(a_rho), t1 = wave_sum(a_q);
        if (lane == 0) {
          bc[0] = t0;
          bc[1] = t1;
        }
      }
    }
    __syncthreads();
    if (sflag[0]) return;
    const double rho = bc[0], q1 = bc[1];
    // ---- decisions (k_pcgs_spmv<0> prologue): test of the previous iteration, rho, beta ------------------------------
    int own_stop = 0, term = 0, res_it = it;
    double beta = 0.0;
    if (need_test) {
      // Q-model test (conjugate_gradient.hpp:239-276); residual-based test is off (r_tolerance = -1)
      const double zeta = it * (q1 - q_prev) / q1;
      if (zeta < P.q_tolerance && it >= P.min_it) {
        own_stop = 1;
        term = 1;
      } else if (it >= P.max_it) {
        own_stop = 1;
        term = 0;
      }
    }
    if (!own_stop) {
      if (rho == 0.0 || isinf(rho) || rho != rho) {
        own_stop = 1;
        term = 2;  // "Numerical failure. rho / beta"
        res_it = it + 1;
      } else if (it > 0) {
        beta = rho / rho_prev;
        if (beta == 0.0 || isinf(beta)) {
          own_stop = 1;
          term = 2;
          res_it = it + 1;
        }
      }
    }
    if (own_stop) {
      finish(term, res_it, 0, false, rho, q1, 0.0, 0.0, 0.0);
      return;
    }
    if (need_test) q_prev = q1;
    // ---- direction p = z + beta p of the staged columns, product, p.q -------------------------------------------------
    if (stager) {
      const S bs = S(beta);
#pragma unroll
      for (int a = 0; a < 9; ++a) {
        const S po = pst[9 * tid + a];
        pst[9 * tid + a] = it == 0 ? zv[a] : zv[a] + bs * po;
      }
    }
    __syncthreads();
    const double qd = product(pst);
    S pc = S(0), qs = S(0);
    double my_pq = 0.0;
    if (rowt) {
      pc = pst[9 * self + ra];
      qs = S(qd);
      qs += lambda * pc;  // pose damping term of right_multiply
      my_pq = double(pc) * double(qs);
    }
    {
      double s0, s1;
      wg_sum2(my_pq, 0.0, s0, s1);
      if (tid == 0) pg_store(P.part_pq, size_t(g), s0, tag);
    }
    // ---- exchange 2: partial sums of p.q -------------------------------------------------------------------------------
    if (gatherer) {
      pg_u32 spins = 0;
      double a_pq = 0.0;
      for (;;) {
        bool ok = true;
        a_pq = 0.0;
#pragma unroll
        for (int k = 0; k < kPgMaxGroups / 64; ++k) {
          const int gg = lane + 64 * k;
          if (gg < P.G) {
            double v0;
            pg_load(P.part_pq, size_t(gg), tag, ok, v0);
            a_pq += v0;
          }
        }
        if (__ballot(!ok) == 0) break;
        if (spin_failed(spins)) {
          raise_abort();
          break;
        }
      }
      const double t0 = wave_sum(a_pq);
      if (lane == 0) bc[2] = t0;
    }
    __syncthreads();
    if (sflag[0]) return;
    const double pq = bc[2];
    // ---- step (k_pcgs_update) ------------------------------------------------------------------------------------------
    {
      int stop2 = 0, term2 = 0;
      double alpha = 0.0;
      if (pq != pq) {
        stop2 = 1;  // NaN: numerical failure at once
        term2 = 2;
      } else if (pq <= 0.0 || isinf(pq)) {
        stop2 = 1;  // "Matrix is indefinite, no more progress can be made." -> NO_CONVERGENCE
      } else {
        alpha = rho / pq;
        if (isinf(alpha)) {
          stop2 = 1;
          term2 = 2;
        }
      }
      if (stop2) {
        // (the state of an iteration that was started: iter = it, cur = it + 1)
        finish(term2, cur, term2 == 0 ? 1 : 0, true, rho, q1, beta, pq, alpha);
        return;
      }
      const S a = S(alpha);
      const bool refresh = (cur % P.period) == 0;
      if (rowt) {
        x_i += a * pc;
        if (!refresh) r_i -= a * qs;
      }
      if (refresh) {
        // residual refresh r = b - H x (conjugate_gradient.hpp:230-235): x travels like z
        if (rowt) pg_store(P.xg, size_t(9) * c + ra, x_i, tag);
        if (wave_stages) {
          pg_u32 spins = 0;
          S xv[9];
#pragma unroll
          for (int a2 = 0; a2 < 9; ++a2) xv[a2] = S(0);
          for (;;) {
            bool ok = true;
            if (stager) {
#pragma unroll
              for (int a2 = 0; a2 < 9; ++a2) pg_load(P.xg, size_t(9) * mycol + a2, tag, ok, xv[a2]);
            }
            if (__ballot(!ok) == 0) break;
            if (spin_failed(spins)) {
              raise_abort();
              break;
            }
          }
          if (stager) {
#pragma unroll
            for (int a2 = 0; a2 < 9; ++a2) opx[9 * tid + a2] = xv[a2];
          }
        }
        __syncthreads();
        if (sflag[0]) return;
        const double q2 = product(opx);
        if (rowt) {
          S q2s = S(q2);
          q2s += lambda * x_i;
          r_i = b_i - q2s;
        }
      }
    }
    rho_prev = rho;
    it = cur;
    need_test = 1;
    close_residual(tag + 1);
  }
}

}  // namespace rba
