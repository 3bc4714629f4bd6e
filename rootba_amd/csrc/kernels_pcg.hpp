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

// HALF STORAGE (the assembled reduced matrix of the square-root solver; round 4). The matrix is symmetric and its
// values are double (kernels_a64.hpp), so the SpMV is a stream of 8-byte scalars. Every off-diagonal block {c, d} is
// stored ONCE, in the row of its OWNER - c or d, chosen on the host so that every row owns about half of its blocks
// and as few rows as possible more than one wavefront's worth (Solver::build_explicit_structure; an upper-triangular
// assignment would give the first rows all of theirs and the last rows none: the longest wavefront decides the kernel
// time) - as the owner sees it (S_cd in row c, S_dc = S_cd^T in row d).
// The wavefront of row c multiplies its block (c, j) twice out of the same LDS copy - S_cj v_j into its own row sums
// and the transposed S_cj^T v_c into a 9-double slot of `tpart` - and the slots of the blocks row j does NOT own are
// laid out row by row, so whoever consumes q_j adds a CONTIGUOUS run of slots in a fixed order (QPieces /
// pcgs_gather_q: bitwise reproducible, identical on all ranks of a sharded run - an atomic scatter would not be).
// p.q needs no complete q: it is the sum over stored blocks of w p_c^T S_cj p_j with w = 2 off the diagonal.
// Half the bytes per product (venice-1778: 35 instead of 70 MB), half the all-reduce of a sharded assembly.
// A camera with very many neighbours it does not own (dense co-visibility: a landmark seen by most cameras connects them
// all) would have one work-item gather hundreds of slots: above kHalfLowerMax (RBA_HALF_LOWER_MAX) the slots of such a HEAVY row are summed by
// a wavefront of its own right behind the product (k_pcgs_reduce_slots) into one more "further item" of the row.
// (Round 4's first form stored such blocks in both rows instead: on a nearly dense matrix - venice with heavy-tailed
// track lengths - that was full storage, 2 GB per product instead of 1 GB.)
constexpr int kHalfLowerMax = 192;

struct HeavyRow {
  int row;    // camera
  int slot0;  // its run of received slots in tpart
  int slot1;
  int extra;  // index of its "further item" in qextra
};

// where the pieces of a product q = M v lie
template <class S>
struct QPieces {
  const S* __restrict__ qmain;         // [9 n_c] sums of a row's first item
  const S* __restrict__ qextra;        // [9 n_extra] sums of the further items of long rows ...
  const int* __restrict__ extra_ptr;   // [n_c + 1]    ... of row c: extra_ptr[c] .. extra_ptr[c + 1]
  const double* __restrict__ tpart;    // [9 n_slots] half storage: transposed contributions, grouped by RECEIVING row
  const int* __restrict__ low_ptr;     // [2 n_c]     (nullptr tpart: full storage) ... row c: low_ptr[2 c] .. low_ptr[2 c + 1]
};

struct SpmvItem {
  int row;    // camera (block row)
  int slot0;  // first block slot of the item
  int slot1;  // one past the last
  int extra;  // -1: first item of its row (writes q[9 row ..], adds lambda p_c, publishes p_new[c]);
              // >= 0: further item of a long row, writes its partial sums to qextra[9 extra ..]
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

constexpr int kSpmvPass = 21;  // 16-byte loads in flight per lane: 21.5 KB per wavefront, ONE chunk of either scalar
// blocks per chunk = what one pass of kSpmvPass loads per lane covers: 64 float blocks, 32 double blocks (a second
// pass would be a second, dependent memory round trip of the wavefront)
template <class S>
constexpr int spmv_chunk_blocks() {
  return sizeof(S) == 8 ? 33 : 64;  // (81 CB scalars + alignment slack <= 64 kSpmvPass 16-byte vectors)
}
template <class S>
constexpr size_t spmv_lds_bytes() {
  return (size_t(spmv_chunk_blocks<S>()) * 81 + Vec16<S>::N) * sizeof(S);
}

// One 64-block chunk of a row strip: the chunk's scalars [81 chunk, 81 (chunk + nb)) are copied to
// LDS at the same 16-byte phase with fully coalesced 16-byte loads (kSpmvPass per lane in flight).
template <class S>
struct ChunkStage {
  using V = typename Vec16<S>::type;
  static constexpr int N = Vec16<S>::N;
  const V* __restrict__ src;
  int off, nvec;
  __device__ __forceinline__ void setup(const S* __restrict__ vals, int chunk, int nb) {
    const int64_t g0 = int64_t(81) * chunk;
    const int64_t base = g0 & ~int64_t(N - 1);
    off = int(g0 - base);
    nvec = (off + 81 * nb + N - 1) / N;
    src = reinterpret_cast<const V*>(vals + base);
  }
  __device__ __forceinline__ void issue(int v0, int lane, V tmp[kSpmvPass]) const {
    // (index clamped instead of predicated: no exec-mask branch per load; surplus lanes re-read
    //  the last vector, the store below is predicated)
#pragma unroll
    for (int u = 0; u < kSpmvPass; ++u) tmp[u] = src[min(v0 + u * 64 + lane, nvec - 1)];
  }
  // every piece, also the lanes past the chunk's end (into the slack of a kSpmvPass KiB slot): straight-line code - the
  // streaming kernel's two register buffers are told apart by the compiler's wait counts, which are exact only there
  __device__ __forceinline__ void store_all(int lane, const V tmp[kSpmvPass], S* lds) const {
    V* dst = reinterpret_cast<V*>(lds);
#pragma unroll
    for (int u = 0; u < kSpmvPass; ++u) dst[u * 64 + lane] = tmp[u];
  }
  __device__ __forceinline__ void store(int v0, int lane, const V tmp[kSpmvPass], S* lds) const {
    V* dst = reinterpret_cast<V*>(lds);
#pragma unroll
    for (int u = 0; u < kSpmvPass; ++u) {
      const int i = v0 + u * 64 + lane;
      if (i < nvec) dst[i] = tmp[u];
    }
  }
};

// lane j multiplies ITS block out of LDS (lane stride 81 words: conflict-free). MT = matrix scalar: a double matrix
// with float vectors (the assembled matrix of a float solver, kernels_a64.hpp) is multiplied in double.
// HALF (see the top of the file): the same pass over the block also forms the transposed product S_cj^T v_c (stored to
// the block's slot `tdst` unless the block is the diagonal one or is stored in both rows) and the lane's share
// w v_c^T S_cj v_j of p.q.
// (fused multiply-adds spelled out: left to the compiler's contraction, which of a float block's products are fused and
//  which become a packed multiply and an add is decided per kernel - the item kernel and the streaming kernels then
//  differ in the last bit of a product)
__device__ __forceinline__ float fma_of(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_of(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <class S, class MT, bool HALF>
__device__ __forceinline__ void spmv_block_times(const MT* lds, int off, int lane, bool act, const S xv[9],
                                                 double acc[9], const S vc[9], bool single, MT tt[9], double& pq) {
#pragma unroll
  for (int b = 0; b < 9; ++b) tt[b] = MT(0);
  if (act) {
    const MT* blk = lds + off + 81 * lane;
    MT dot = MT(0);
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      // (three block rows at a time: left alone the scheduler hoists all 81 LDS reads of a double block - 162
      //  registers - above the arithmetic, and the kernel needs more than 256 registers: one wavefront per SIMD)
      if (sizeof(MT) == 8 && a % 3 == 0 && a > 0) __builtin_amdgcn_sched_barrier(0);
      MT t = MT(0);
#pragma unroll
      for (int b = 0; b < 9; ++b) {
        const MT v = blk[9 * a + b];
        t = fma_of(v, MT(xv[b]), t);
        if (HALF) tt[b] = fma_of(v, MT(vc[a]), tt[b]);
      }
      acc[a] += double(t);
      if (HALF) dot = fma_of(MT(vc[a]), t, dot);
    }
    if (HALF) pq += double(dot) * (single ? 2.0 : 1.0);
  }
}

// The transposed products of a chunk's blocks to their slots (half storage; td: the lane's slot, -1 = none), WHOLE SLOTS BY
// NEIGHBOURING LANES: the lanes' nine values are laid side by side in LDS - the chunk's slot, whose multiply is over - and
// stored as one run of 9 nb doubles, lane by lane, so that the 72 bytes of a slot leave in one memory operation of nine
// neighbouring lanes. Stored by its own lane a slot is five partial writes (four 16-byte pieces and one of 8 bytes,
// each a transaction of its own at the L2: 165 per chunk of 33 blocks - as many as the 168 lines of the chunk's
// matrix bytes). Worth 3 % of the product of the double matrix of venice-1778+tail (stream 96.7 / 94.1 -> 91.9 / 92.5 us,
// item kernel 91.6 / 94.7 -> 90.0 / 90.9; profiles/r6aa_*) - NOT the 16 us the slot stores cost in all (timing-only
// variant without them: 79 us, profiles/r6z_*): that is their 56 MB of partial-line writes among 383 MB of reads, not
// their number of transactions. It also takes nine 64-bit store addresses per lane out of the multiply: the double kernels
// need 130-160 registers instead of 202-219, and the streaming form with one chunk in flight fits two wavefronts per
// SIMD for a double matrix too. Called by all 64 lanes.
template <class MT, int CB>
__device__ __forceinline__ void store_slots(MT* lds, int lane, int nb, const MT tt[9], int td,
                                            double* __restrict__ tpart) {
  if constexpr (sizeof(MT) == 4) {
    // (a float matrix - 64 blocks per chunk, nine passes below - measured 4 % SLOWER that way: every lane its own slot)
    if (lane < nb && td >= 0) {
#pragma unroll
      for (int b = 0; b < 9; ++b) tpart[size_t(9) * td + b] = double(tt[b]);
    }
    return;
  }
  MT* sv = lds;
  int* st = reinterpret_cast<int*>(lds + 9 * CB);
  wave_lds_fence();  // (the multiply has read the chunk)
  if (lane < nb) {
#pragma unroll
    for (int b = 0; b < 9; ++b) sv[9 * lane + b] = tt[b];
    st[lane] = td;
  }
  wave_lds_fence();
  constexpr int PASSES = (9 * CB + 63) / 64;
#pragma unroll
  for (int u = 0; u < PASSES; ++u) {
    const int e = 64 * u + lane;
    if (e < 9 * nb) {
      const int sl = e / 9;
      const int t = st[sl];
      if (t >= 0) tpart[size_t(9) * t + (e - 9 * sl)] = double(sv[e]);
    }
  }
}

// MODE 0: direction update + product + p.q partial.   MODE 1: refresh product S x (+ lambda x).
// MODE 2: plain product S x + lambda x outside the fused loop (series of the power preconditioner): lambda is
//         passed in `q_tolerance`, only `done` is read from the state.
// Load schedule (the kernel is a chain of memory round trips, not a bandwidth problem):
//   round 1  the item descriptor
//   round 2  the first chunk's matrix vectors, its column indices, the 2 x 64 reduction partials and
//            the PCG state - all independent, all issued before anything is waited for
//   round 3  the gathers of z / p (need the column indices)
// The termination decision is evaluated while rounds 2/3 are in flight.
template <class S, int MODE, class MT = S, bool HALF = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_pcgs_spmv(const int* __restrict__ cols, const MT* __restrict__ vals,
                                                  const SpmvItem* __restrict__ items, const S* __restrict__ z,
                                                  S* pbuf0, S* pbuf1, const S* __restrict__ xvec,
                                                  S* __restrict__ qmain, S* __restrict__ qextra,
                                                  double* __restrict__ tpart, const int* __restrict__ tdst,
                                                  CgState* st, const double* __restrict__ part_rho,
                                                  const double* __restrict__ part_q,
                                                  double* __restrict__ part_pq, double q_tolerance, int min_it,
                                                  int max_it, int period, int* host_progress) {
  using V = typename Vec16<MT>::type;
  extern __shared__ __attribute__((aligned(16))) char smem_pcgs[];
  MT* lds = reinterpret_cast<MT*>(smem_pcgs);
  const int lane = threadIdx.x;
  const SpmvItem item = items[blockIdx.x];
  const int c = item.row;
  // ---- round 2: everything that depends on the item only. vmcnt retires in order: what the
  //      next round needs (column indices, partials) is requested BEFORE the 21 matrix vectors
  ChunkStage<MT> cs;
  constexpr int CB = spmv_chunk_blocks<MT>();
  static_assert((81 * CB + 2 * (Vec16<MT>::N - 1)) / Vec16<MT>::N <= 64 * kSpmvPass, "a chunk is one pass of loads");
  const int nb0 = min(CB, item.slot1 - item.slot0);
  const bool act0 = lane < nb0;
  const int col0 = cols[item.slot0 + min(lane, nb0 - 1)];
  const int td0 = HALF ? tdst[item.slot0 + min(lane, nb0 - 1)] : -1;  // slot of the transposed product (-1: none)
  double prho = 0, pq1 = 0;
  if (MODE == 0) {
    prho = part_rho[lane];
    pq1 = part_q[lane];
  }
  const int done = st->done, it = st->iter, cur_st = st->cur, need_test = st->need_test, pswap = st->pswap;
  const double q_prev = st->q_hist[(it + 1) & 1], rho_prev = st->rho_hist[(it + 1) & 1];
  // (MODE 2: lambda in `q_tolerance`; negative = take it from the device state like the other modes - inside the
  //  captured launch graphs of the PCG, whose arguments are fixed)
  const S lambda = (MODE == 2 && q_tolerance >= 0.0) ? S(q_tolerance) : S(st->lambda);
  cs.setup(vals, item.slot0, nb0);
  V tmp[kSpmvPass];
  cs.issue(0, lane, tmp);
  // ---- round 3: the operand of the first chunk ---------------------------------------------
  const S* __restrict__ p_old = ((it + pswap) & 1) ? pbuf1 : pbuf0;
  S* __restrict__ p_new = ((it + pswap) & 1) ? pbuf0 : pbuf1;
  const bool first_it = it == 0;
  S za[9], pa[9];
  // (unpredicated: col0 is a valid column for every lane; idle lanes are masked at the product)
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if (MODE == 0) {
      za[t] = z[9 * col0 + t];
      pa[t] = p_old[9 * col0 + t];
    } else {
      za[t] = xvec[9 * col0 + t];
    }
  }
  const int lc = min(lane, 8);
  const S zc = MODE == 0 ? z[9 * c + lc] : xvec[9 * c + lc];
  const S pcold = MODE == 0 ? p_old[9 * c + lc] : S(0);

  // ---- decisions (while the loads are in flight) ---------------------------------------------
  int stop = done, term = 0, res_it = it, own_stop = 0;
  double beta = 0.0, rho = 0.0, q1 = 0.0;
  if (MODE == 0) {
    rho = wave_sum(prho);
    q1 = wave_sum(pq1);
    if (!done) {
      if (need_test) {
        // Q-model test (conjugate_gradient.hpp:239-276); residual-based test is off (r_tolerance = -1)
        const double zeta = it * (q1 - q_prev) / q1;
        if (zeta < q_tolerance && it >= min_it) {
          own_stop = 1;
          term = 1;
        } else if (it >= max_it) {
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
    }
    stop = done | own_stop;
    if (blockIdx.x == 0 && lane == 0) {
      if (done) {
        // (the host's run-ahead throttle must learn about a termination the round-1 kernels detected)
        if (host_progress) __hip_atomic_store(host_progress + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        if (need_test) st->q_hist[it & 1] = q1;
        if (own_stop) {
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
    }
  } else if (MODE == 1) {
    stop = done | (cur_st % period != 0 ? 1 : 0);
  } else {
    stop = done;
  }
  if (stop) return;

  // ---- q_c = sum_j S_cj v_j,  v = z + beta p (MODE 0) or x (MODE 1) ---------------------------
  const S bs = S(beta);
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  S xv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) xv[t] = (MODE == 0 && !first_it) ? za[t] + bs * pa[t] : za[t];
  (void)pa;
  const S pc = (MODE == 0 && !first_it) ? zc + bs * pcold : zc;  // lane t < 9: entry t of p_c (MODE 0) / x_c
  S vc[9];  // half storage: v_c in every lane
#pragma unroll
  for (int t = 0; t < 9; ++t) vc[t] = HALF ? read_lane(pc, t) : S(0);
  double pq = 0.0;
  cs.store(0, lane, tmp, lds);
  __syncthreads();
  MT tt[9];
  spmv_block_times<S, MT, HALF>(lds, cs.off, lane, act0, xv, acc, vc, HALF && td0 >= 0, tt, pq);
  if (HALF) store_slots<MT, CB>(lds, lane, nb0, tt, td0, tpart);
  // (half storage: an item is ONE chunk - Solver::build_explicit_structure splits rows at spmv_chunk_blocks<double>())
  for (int chunk = item.slot0 + CB; !HALF && chunk < item.slot1; chunk += CB) {  // long rows only
    __syncthreads();  // the staging buffer is overwritten
    const int nb = min(CB, item.slot1 - chunk);
    const bool act = lane < nb;
    const int col = act ? cols[chunk + lane] : 0;
    const int td = (HALF && act) ? tdst[chunk + lane] : -1;
    cs.setup(vals, chunk, nb);
    cs.issue(0, lane, tmp);
    cs.store(0, lane, tmp, lds);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (MODE == 0) {
        const S zz = act ? z[9 * col + t] : S(0);
        const S pp = (act && !first_it) ? p_old[9 * col + t] : S(0);
        xv[t] = first_it ? zz : zz + bs * pp;
      } else {
        xv[t] = act ? xvec[9 * col + t] : S(0);
      }
    }
    __syncthreads();
    spmv_block_times<S, MT, HALF>(lds, cs.off, lane, act, xv, acc, vc, HALF && td >= 0, tt, pq);  // (full storage only)
  }
  S mine = S(0);
#pragma unroll
  for (int a = 0; a < 9; ++a) {
    const double tot = wave_sum(acc[a]);
    if (lane == a) mine = S(tot);
  }
  if (lane < 9) {
    if (item.extra < 0) {
      mine += lambda * pc;  // pose damping term of right_multiply
      if (MODE == 0) p_new[9 * c + lane] = pc;
      qmain[9 * c + lane] = mine;
      if (HALF) pq += double(lambda) * double(pc) * double(pc);
    } else {
      qextra[9 * item.extra + lane] = mine;
    }
    if (!HALF) pq = double(pc) * double(mine);
  }
  if (MODE == 0) {
    pq = wave_sum(pq);
    if (lane == 0) part_pq[blockIdx.x] = pq;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same product as a STREAM (round 6): half storage, one chunk per item, PERSISTENT
// single-wavefront workgroups - one per SIMD, the whole 512-entry register file each - that walk the items
// w, w + W, w + 2 W, ... with TWO chunks in flight per wavefront at all times.
//
// k_pcgs_spmv above is one short-lived wavefront per item: item descriptor -> (column indices | matrix chunk into 84
// staging registers) -> operand gather -> multiply -> exit, i.e. three dependent memory round trips plus the launch of a
// workgroup around 21.5 KB of matrix, with the chunk in flight for one of the three: seven wavefronts per compute unit
// (the staging buffer) hold at most 150 KB in flight and on average about half of it - 4.5 TB/s on a nearly dense matrix
// (424 MB in 94 us on venice-1778+tail, profiles/r6e_*). It does not matter for a banded matrix (venice: 36 MB, one wave
// of workgroups, and long solves run in the register files anyway); it is the whole iteration for one that does not fit.
// What bounds such a stream is the number of bytes in flight, and the largest memory of a compute unit is its register
// file (512 KB; LDS: 160 KB). Measured first, and dropped: a two-slot LDS ring filled by LDS-DMA (global_load_lds_dwordx4,
// no staging registers) with one wait per chunk - one chunk in flight per wavefront, three wavefronts per unit (43 KB
// of LDS each): 119 us, SLOWER than the item kernel (profiles/r6e_tail_kernel_stats_stream_lds_dma.csv); hipcc waits for
// ALL outstanding memory operations while a DMA is in flight, so a deeper DMA ring needs every other load in assembly too.
// Here: chunks k + 1 and k + 2 are in flight in two register buffers (2 x 84) while chunk k is multiplied out of the
// wavefront's LDS slot; the operand entries of chunk k + 1 are gathered, the column indices of chunk k + 3 and the
// descriptor of chunk k + 4 read, one iteration before they are needed, and requested BEFORE the matrix loads of the
// same iteration (loads return in order: what the next multiply waits for must not queue behind a chunk that is needed
// two iterations later). Same arithmetic, same summation order per item as k_pcgs_spmv: bit-identical products
// (tests/test_gpu_parity.py::test_streaming_spmv_is_the_item_spmv).
// What the measurements at the end of round 6 said: bytes in flight are NOT the bound - reading the
// 424-MB matrix through this kernel takes 79 us (5.4 TB/s, the rate of the library's other read-bound kernels), the 56 MB
// of transposed-product slots cost 13-17 us more. The form with ONE register buffer (NB = 1: chunk k + 1 in flight
// while chunk k is multiplied) and two wavefronts per SIMD - seven per compute unit, their LDS slots - is as fast on the
// double matrix and 9 % faster on the float copy of final-13682: it is the default (k_pcgs_spmv_stream1;
// RBA_SPMV_STREAM_BUFFERS=2 selects the two-buffer form).
// the nine entries of a camera in three memory operations (4-byte aligned 16-byte loads + one scalar; exactly the
// 36 bytes: a three-component vector type is sixteen bytes wide on the host side of the test harness)
__device__ __forceinline__ void load_nine(const float* __restrict__ p, float (&v)[9]) {
  typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
  const f4 a = *reinterpret_cast<const f4*>(p), b = *reinterpret_cast<const f4*>(p + 4);
  v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w;
  v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
  v[8] = p[8];
}
__device__ __forceinline__ void load_nine(const double* __restrict__ p, double (&v)[9]) {
  typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const d2 t = *reinterpret_cast<const d2*>(p + 2 * u);
    v[2 * u] = t.x;
    v[2 * u + 1] = t.y;
  }
  v[8] = p[8];
}

template <class S, int MODE, class MT, int NB>
__device__ __forceinline__ void pcgs_spmv_stream_body(
    const int* __restrict__ cols, const MT* __restrict__ vals, const SpmvItem* __restrict__ items, int n_items,
    const S* __restrict__ z, S* pbuf0, S* pbuf1, const S* __restrict__ xvec, S* __restrict__ qmain,
    S* __restrict__ qextra, double* __restrict__ tpart, const int* __restrict__ tdst, CgState* st,
    const double* __restrict__ part_rho, const double* __restrict__ part_q, double* __restrict__ part_pq,
    double q_tolerance, int min_it, int max_it, int period, int* host_progress) {
  using V = typename Vec16<MT>::type;
  extern __shared__ __attribute__((aligned(16))) char smem_pcgs[];
  MT* lds = reinterpret_cast<MT*>(smem_pcgs);
  const int lane = threadIdx.x, W = int(gridDim.x);
  constexpr int CB = spmv_chunk_blocks<MT>();
  // ---- prologue: the decisions of k_pcgs_spmv, by every wavefront alike ---------------------------------------------
  double prho = 0, pq1 = 0;
  if (MODE == 0) {
    prho = part_rho[lane];
    pq1 = part_q[lane];
  }
  const int done = st->done, it = st->iter, cur_st = st->cur, need_test = st->need_test, pswap = st->pswap;
  const double q_prev = st->q_hist[(it + 1) & 1], rho_prev = st->rho_hist[(it + 1) & 1];
  const S lambda = (MODE == 2 && q_tolerance >= 0.0) ? S(q_tolerance) : S(st->lambda);
  int k = int(blockIdx.x);
  // (descriptors of the first four items: independent of the decisions, requested before them. As four-integer vectors
  //  {row, slot0, slot1, extra} in registers: a SpmvItem selected between two sources is a stack object, and a scratch
  //  load in the loop is a wait for EVERYTHING in flight)
  static_assert(sizeof(SpmvItem) == sizeof(int4), "descriptor = one 16-byte load");
  const int4* __restrict__ item_vec = reinterpret_cast<const int4*>(items);
  int4 i0 = item_vec[min(k, n_items - 1)], i1 = item_vec[min(k + W, n_items - 1)];
  int4 i2 = item_vec[min(k + 2 * W, n_items - 1)], i3 = item_vec[min(k + 3 * W, n_items - 1)];
  int stop = done, term = 0, res_it = it, own_stop = 0;
  double beta = 0.0, rho = 0.0, q1 = 0.0;
  if (MODE == 0) {
    rho = wave_sum(prho);
    q1 = wave_sum(pq1);
    if (!done) {
      if (need_test) {
        const double zeta = it * (q1 - q_prev) / q1;
        if (zeta < q_tolerance && it >= min_it) {
          own_stop = 1;
          term = 1;
        } else if (it >= max_it) {
          own_stop = 1;
          term = 0;
        }
      }
      if (!own_stop) {
        if (rho == 0.0 || isinf(rho) || rho != rho) {
          own_stop = 1;
          term = 2;
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
    }
    stop = done | own_stop;
    if (blockIdx.x == 0 && lane == 0) {
      if (done) {
        if (host_progress) __hip_atomic_store(host_progress + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        if (need_test) st->q_hist[it & 1] = q1;
        if (own_stop) {
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
    }
  } else if (MODE == 1) {
    stop = done | (cur_st % period != 0 ? 1 : 0);
  }
  if (stop || k >= n_items) return;
  const S* __restrict__ p_old = ((it + pswap) & 1) ? pbuf1 : pbuf0;
  S* __restrict__ p_new = ((it + pswap) & 1) ? pbuf0 : pbuf1;
  const bool comb = MODE == 0 && it != 0;  // v = z + beta p_old (else z, or x)
  const S* __restrict__ v0 = MODE == 0 ? z : xvec;
  const S bs = S(beta);
  const int lc = min(lane, 8);
  auto blocks_of = [&](const int4& i) { return min(CB, i.z - i.y); };
  // ---- fill: column indices of chunks k, k + W, k + 2 W; operand of chunk k; chunks k and k + W on their way ---------
  int nb0 = blocks_of(i0), nb1 = blocks_of(i1), nb2 = blocks_of(i2);
  int c0 = cols[i0.y + min(lane, nb0 - 1)], t0 = tdst[i0.y + min(lane, nb0 - 1)];
  int c1 = cols[i1.y + min(lane, nb1 - 1)], t1 = tdst[i1.y + min(lane, nb1 - 1)];
  int c2 = cols[i2.y + min(lane, nb2 - 1)], t2 = tdst[i2.y + min(lane, nb2 - 1)];
  S xv[9], xp[9];
  load_nine(v0 + 9 * c0, xv);
  if (comb) {
    load_nine(p_old + 9 * c0, xp);
  } else {
#pragma unroll
    for (int t = 0; t < 9; ++t) xp[t] = S(0);
  }
  S zc = v0[9 * i0.x + lc], pcold = comb ? p_old[9 * i0.x + lc] : S(0);
  __builtin_amdgcn_sched_barrier(0);
  V bufA[kSpmvPass], bufB[kSpmvPass];
  ChunkStage<MT> csA, csB;
  csA.setup(vals, i0.y, nb0);
  csA.issue(0, lane, bufA);
  __builtin_amdgcn_sched_barrier(0);  // (A before B here too: the wait counts at the loop head are the minimum over its two entries)
  if (NB == 2) {
    csB.setup(vals, i1.y, nb1);
    csB.issue(0, lane, bufB);  // (a wavefront with one item re-reads it: clamped descriptors, never stored)
  }
  __builtin_amdgcn_sched_barrier(0);

  // one iteration: chunk k arrives in `buf` (stage `cs`), goes to LDS and is multiplied; `buf` is refilled with chunk
  // k + 2 W. Returns false behind the wavefront's last chunk.
  // (straight-line code, ONE copy per buffer: with exact wait counts the wait for one buffer does not include the other's
  //  loads. The refill is unconditional - behind the wavefront's last item the clamped descriptor names the matrix's
  //  last chunk, which all wavefronts then re-read from L2 and never store; copies of this step without the request for
  //  the last two chunks were merged with the loop's by the compiler, and the loop head then waited for everything)
  auto step = [&](V(&buf)[kSpmvPass], ChunkStage<MT>& cs) __attribute__((always_inline)) {
    cs.store_all(lane, buf, lds);
    const int off = cs.off;
    __builtin_amdgcn_sched_barrier(0);
    // ---- requests of the next iterations: the small ones first, then the matrix chunk k + 2 W. Unconditional (clamped
    //      descriptors: valid addresses behind the wavefront's last item) and fenced against the instruction scheduler: loads
    //      return in order, so a small load queued behind the 21 pieces of the chunk would make the next multiply wait for
    //      a chunk that is needed two iterations later. (A wavefront keeps at most 63 memory operations outstanding: nine
    //      entries of a camera are three 12-byte loads, not nine.)
    S xv1[9], xp1[9];
    load_nine(v0 + 9 * c1, xv1);
    if (comb) {
      load_nine(p_old + 9 * c1, xp1);
    } else {
#pragma unroll
      for (int t = 0; t < 9; ++t) xp1[t] = S(0);
    }
    const S zc1 = v0[9 * i1.x + lc], pcold1 = comb ? p_old[9 * i1.x + lc] : S(0);
    const int nb3 = blocks_of(i3);
    const int c3 = cols[i3.y + min(lane, nb3 - 1)], t3 = tdst[i3.y + min(lane, nb3 - 1)];
    const int4 i4 = item_vec[min(k + 4 * W, n_items - 1)];
    __builtin_amdgcn_sched_barrier(0);
    if (NB == 2) cs.setup(vals, i2.y, nb2);
    else cs.setup(vals, i1.y, nb1);
    cs.issue(0, lane, buf);
    __builtin_amdgcn_sched_barrier(0);
    // ---- multiply chunk k out of LDS ----------------------------------------------------------------------------------
    wave_lds_fence();
    const bool act = lane < nb0;
#pragma unroll
    for (int t = 0; t < 9; ++t) xv[t] = comb ? xv[t] + bs * xp[t] : xv[t];
    const S pc = comb ? zc + bs * pcold : zc;  // lane t < 9: entry t of p_c (MODE 0) / x_c
    S vc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) vc[t] = read_lane(pc, t);
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, pq = 0.0;
    MT tt[9];
    spmv_block_times<S, MT, true>(lds, off, lane, act, xv, acc, vc, t0 >= 0, tt, pq);
    store_slots<MT, CB>(lds, lane, nb0, tt, t0, tpart);
    S mine = S(0);
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      const double tot = wave_sum(acc[a]);
      if (lane == a) mine = S(tot);
    }
    if (lane < 9) {
      if (i0.w < 0) {
        mine += lambda * pc;
        if (MODE == 0) p_new[9 * i0.x + lane] = pc;
        qmain[9 * i0.x + lane] = mine;
        pq += double(lambda) * double(pc) * double(pc);
      } else {
        qextra[9 * i0.w + lane] = mine;
      }
    }
    if (MODE == 0) {
      pq = wave_sum(pq);
      if (lane == 0) part_pq[k] = pq;
    }
    wave_lds_fence();  // (the next chunk overwrites the slot)
    // ---- chunk k + W becomes the current one ----------------------------------------------------------------------------
    k += W;
    i0 = i1;
    i1 = i2;
    i2 = i3;
    i3 = i4;
    nb0 = nb1;
    nb1 = nb2;
    nb2 = nb3;
    c0 = c1;
    c1 = c2;
    c2 = c3;
    t0 = t1;
    t1 = t2;
    t2 = t3;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      xv[t] = xv1[t];
      xp[t] = xp1[t];
    }
    zc = zc1;
    pcold = pcold1;
  };
  if (NB == 2) {
    for (;;) {
      step(bufA, csA);
      if (k >= n_items) break;
      step(bufB, csB);
      if (k >= n_items) break;
    }
  } else {
    do step(bufA, csA);
    while (k < n_items);
  }
}

#define RBA_SPMV_STREAM_ARGS                                                                                      \
  const int *__restrict__ cols, const MT *__restrict__ vals, const SpmvItem *__restrict__ items, int n_items,     \
      const S *__restrict__ z, S *pbuf0, S *pbuf1, const S *__restrict__ xvec, S *__restrict__ qmain,             \
      S *__restrict__ qextra, double *__restrict__ tpart, const int *__restrict__ tdst, CgState *st,              \
      const double *__restrict__ part_rho, const double *__restrict__ part_q, double *__restrict__ part_pq,       \
      double q_tolerance, int min_it, int max_it, int period, int *host_progress
#define RBA_SPMV_STREAM_PASS                                                                                      \
  cols, vals, items, n_items, z, pbuf0, pbuf1, xvec, qmain, qextra, tpart, tdst, st, part_rho, part_q, part_pq,   \
      q_tolerance, min_it, max_it, period, host_progress
// two chunks in flight per wavefront, one wavefront per SIMD
template <class S, int MODE, class MT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_pcgs_spmv_stream(RBA_SPMV_STREAM_ARGS) {
  pcgs_spmv_stream_body<S, MODE, MT, 2>(RBA_SPMV_STREAM_PASS);
}
// one chunk in flight per wavefront, two wavefronts per SIMD (as many as LDS slots fit on a compute unit: seven)
template <class S, int MODE, class MT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_pcgs_spmv_stream1(RBA_SPMV_STREAM_ARGS) {
  pcgs_spmv_stream_body<S, MODE, MT, 1>(RBA_SPMV_STREAM_PASS);
}
#undef RBA_SPMV_STREAM_ARGS
#undef RBA_SPMV_STREAM_PASS

// float copy of the assembled double matrix (the terms of the power-series preconditioner: Solver::series_f32)
__global__ __launch_bounds__(256) void k_narrow_matrix(const double* __restrict__ src, float* __restrict__ dst, size_t n) {
  const size_t stride = size_t(gridDim.x) * 256 * 4;
  for (size_t i = (size_t(blockIdx.x) * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      const double4 v = *reinterpret_cast<const double4*>(src + i);
      *reinterpret_cast<float4*>(dst + i) = float4{float(v.x), float(v.y), float(v.z), float(v.w)};
    } else {
      for (size_t j = i; j < n; ++j) dst[j] = float(src[j]);
    }
  }
}

// q_i = first item's sum + the extra items of a long row + (half storage) the transposed contributions of the blocks
// other rows own, each in a fixed order. The slots of a row are contiguous: all loads of a batch of sixteen are
// requested together (one memory round trip per batch).
template <class S>
__device__ __forceinline__ S pcgs_gather_q(const QPieces<S>& qp, S qm, int e0, int e1, int l0, int l1, int row) {
  for (int q = e0; q < e1; ++q) qm += qp.qextra[9 * q + row];
  if (qp.tpart == nullptr || l1 <= l0) return qm;
  double acc = double(qm);
  for (int base = l0; base < l1; base += 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = qp.tpart[size_t(9) * min(base + u, l1 - 1) + row] * (base + u < l1 ? 1.0 : 0.0);
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];
  }
  return S(acc);
}
template <class S>
__device__ __forceinline__ S pcgs_gather_q(const QPieces<S>& qp, int c, int row) {
  const bool half = qp.tpart != nullptr;
  return pcgs_gather_q(qp, qp.qmain[9 * c + row], qp.extra_ptr[c], qp.extra_ptr[c + 1], half ? qp.low_ptr[2 * c] : 0,
                       half ? qp.low_ptr[2 * c + 1] : 0, row);
}

// phase 0: after the direction product; phase 1: after the refresh product.
// All operands of the first tile are requested before the p.q reduction is waited for.
template <class S>
__global__ __launch_bounds__(256) void k_pcgs_update(const S* __restrict__ inv, const S* __restrict__ bvec,
                                                     S* __restrict__ x, S* __restrict__ r, S* __restrict__ z,
                                                     const S* pbuf0, const S* pbuf1, QPieces<S> qp, int n_items,
                                                     int n_cams,
                                                     CgState* st, const double* __restrict__ part_pq,
                                                     double* __restrict__ part_rho, double* __restrict__ part_q,
                                                     int phase, int period, int* host_progress, int mf, S lambda_mf,
                                                     S* __restrict__ zero_me, S* __restrict__ tser) {
  // `tser`: power-series preconditioner - z = Hpp^-1 r is only the first term: it is also the first `t` of the series
  // (k_pcgs_series_step adds the others and then replaces the partials of rho this kernel leaves).
  // `mf`: the product came from the matrix-free operator (k_hx_implicit*: `qmain` = sum_l A_l^T A_l v without the pose
  // damping, no item partials): p.q is summed here, by every workgroup alike, over the whole vectors, and lambda_mf v is
  // added where the product is used. `zero_me`: the accumulator of the refresh product, cleared on refresh iterations.
  __shared__ double sm[4][2];
  __shared__ S rl[252];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int n_tiles = (n_cams + 27) / 28;
  // ---- loads first --------------------------------------------------------------------------
  double accp = 0;
  if (phase == 0 && mf) {
    const int n = 9 * n_cams;
    const S* __restrict__ pv = ((st->cur + st->pswap) & 1) ? pbuf1 : pbuf0;
    for (int base = 0; base < n; base += 2048) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        // (clamped and masked, not selected: a value that is only used under `idx < n` gets its load sunk into a
        //  conditional block, and the sixteen loads of a batch become sixteen waits)
        const int idx = base + u * 256 + tid, ic = min(idx, n - 1);
        const S pi = pv[ic], qi = qp.qmain[ic] + lambda_mf * pi;
        v[u] = double(pi) * double(qi) * (idx < n ? 1.0 : 0.0);
      }
      accp += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
  } else if (phase == 0) {
    // eight independent loads per thread and batch (a serial strided loop would be eight memory
    // round trips); fixed summation order
    for (int base = 0; base < n_items; base += 2048) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256 + tid;
        v[u] = part_pq[min(idx, n_items - 1)];
        if (idx >= n_items) v[u] = 0.0;
      }
      accp += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
  }
  const int done = st->done, cur = st->cur, pswap = st->pswap;
  const double rho0 = st->rho_hist[0], rho1 = st->rho_hist[1];
  int tile = blockIdx.x;
  int i = 252 * tile + tid;
  bool act = tile < n_tiles && tid < 252 && i < 9 * n_cams;
  int c = act ? i / 9 : 0, row = act ? i - 9 * c : 0;
  S xo = S(0), ro = S(0), bo = S(0), po0 = S(0), po1 = S(0), qm = S(0), Mrow[9];
  int e0 = 0, e1 = 0, l0 = 0, l1 = 0;
  const bool half = qp.tpart != nullptr;
#pragma unroll
  for (int j = 0; j < 9; ++j) Mrow[j] = S(0);
  if (act) {
    xo = x[i];
    ro = r[i];
    bo = bvec[i];
    po0 = pbuf0[i];
    po1 = pbuf1[i];
    qm = qp.qmain[i];
    if (!mf) {
      e0 = qp.extra_ptr[c];
      e1 = qp.extra_ptr[c + 1];
      if (half) {
        l0 = qp.low_ptr[2 * c];
        l1 = qp.low_ptr[2 * c + 1];
      }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) Mrow[j] = inv[81 * c + 9 * row + j];
  }
  // the product's pieces of the first tile are gathered NOW: the loads (a second round trip behind the pointers above)
  // are in flight while p.q is reduced, instead of a third round trip behind alpha
  S qg = qm;
  if (act && !mf) qg = pcgs_gather_q(qp, qm, e0, e1, l0, l1, row);
  // ---- p.q, alpha ---------------------------------------------------------------------------
  const bool refresh = (cur % period) == 0;
  int stop = done | ((phase == 1 && !refresh) ? 1 : 0);
  S a = S(0);
  if (phase == 0) {
    const double t = wave_sum(accp);
    if (lane == 0) sm[wave][0] = t;
    __syncthreads();
    const double pq = (sm[0][0] + sm[1][0]) + (sm[2][0] + sm[3][0]);
    __syncthreads();
    int own_stop = 0, term = 0;
    double alpha = 0;
    if (pq != pq) {
      own_stop = 1;  // NaN: numerical failure at once (the reference would iterate on NaNs up to max_iterations and the
      term = 2;      // LM loop reject the non-finite increment all the same)
    } else if (pq <= 0.0 || isinf(pq)) {
      own_stop = 1;  // "Matrix is indefinite, no more progress can be made." -> NO_CONVERGENCE
    } else {
      alpha = ((cur + 1) & 1 ? rho1 : rho0) / pq;
      if (isinf(alpha)) {
        own_stop = 1;
        term = 2;
      }
    }
    if (!done && blockIdx.x == 0 && tid == 0) {
      st->pq = pq;
      st->alpha = alpha;
      if (own_stop) {
        st->termination = term;
        st->indefinite = term == 0 ? 1 : 0;
        st->result_iter = cur;
        st->done = 1;
        if (host_progress) __hip_atomic_store(host_progress + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    stop |= own_stop;
    a = S(alpha);
  }
  if (stop) return;
  const bool odd = (cur + pswap) & 1;
  const S* __restrict__ p = odd ? pbuf1 : pbuf0;
  double acc_rho = 0, acc_q = 0;
  for (; tile < n_tiles; tile += gridDim.x) {
    if (tile != int(blockIdx.x)) {  // further tiles (n_cams > 28 * gridDim.x)
      i = 252 * tile + tid;
      act = tid < 252 && i < 9 * n_cams;
      c = act ? i / 9 : 0;
      row = act ? i - 9 * c : 0;
      if (act) {
        xo = x[i];
        ro = r[i];
        bo = bvec[i];
        po0 = p[i];
        po1 = po0;
        qm = qp.qmain[i];
        if (!mf) {
          e0 = qp.extra_ptr[c];
          e1 = qp.extra_ptr[c + 1];
          if (half) {
            l0 = qp.low_ptr[2 * c];
            l1 = qp.low_ptr[2 * c + 1];
          }
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) Mrow[j] = inv[81 * c + 9 * row + j];
        qg = mf ? qm : pcgs_gather_q(qp, qm, e0, e1, l0, l1, row);
      }
    }
    S xn = S(0), rn = S(0);
    if (act) {
      const S pi = odd ? po1 : po0;
      const S qv = mf ? qm + lambda_mf * (phase == 0 ? pi : xo) : qg;
      if (phase == 0) {
        xn = xo + a * pi;
        x[i] = xn;
        if (!refresh) {
          rn = ro - a * qv;
          r[i] = rn;
        } else if (zero_me) {
          zero_me[i] = S(0);
        }
      } else {
        xn = xo;
        rn = bo - qv;  // qv = (S x + lambda x)_i
        r[i] = rn;
      }
    }
    if (phase == 0 && refresh) continue;  // uniform: the residual comes from the refresh product
    __syncthreads();
    if (tid < 252) rl[tid] = rn;
    __syncthreads();
    if (act) {
      const S* rc = rl + 9 * (tid / 9);
      S zc = S(0);
#pragma unroll
      for (int j = 0; j < 9; ++j) zc += Mrow[j] * rc[j];
      z[i] = zc;
      if (tser) tser[i] = zc;
      acc_rho += double(rn) * double(zc);
      acc_q -= double(xn) * double(bo + rn);
    }
  }
  if (phase == 0 && refresh) return;
  const double t0 = wave_sum(acc_rho), t1 = wave_sum(acc_q);
  if (lane == 0) {
    sm[wave][0] = t0;
    sm[wave][1] = t1;
  }
  __syncthreads();
  if (tid == 0) {
    part_rho[blockIdx.x] = (sm[0][0] + sm[1][0]) + (sm[2][0] + sm[3][0]);
    part_q[blockIdx.x] = (sm[0][1] + sm[1][1]) + (sm[2][1] + sm[3][1]);
    if (blockIdx.x == 0) {
      st->iter = cur;
      st->need_test = 1;
    }
  }
}

// The matrix-free iterations of a solve (before the assembled matrix pays off, or all of them) in the same protocol:
// this kernel is the prologue of k_pcgs_spmv<0> on its own - Q-model test of the previous iteration, rho, beta, state
// hand-over, host progress - followed by the direction update p = z + beta p (in place), the pre-scaled operand D p of
// the product and the cleared accumulator; then k_hx_implicit*, then k_pcgs_update with `mf`. Two vector kernels per
// iteration instead of the five of round 1 (k_pcg_a1 a2 b1 b2 fin, still used by the power-series preconditioner).
template <class S>
__global__ __launch_bounds__(256) void k_pcgs_direction(const S* __restrict__ z, S* __restrict__ pvec, S* __restrict__ q,
                                                        int n, CgState* st, const double* __restrict__ part_rho,
                                                        const double* __restrict__ part_q, const S* __restrict__ dscale,
                                                        S* __restrict__ pscaled, double q_tolerance, int min_it,
                                                        int max_it, int* host_progress, int test_only) {
  // `test_only`: the decisions alone - the host asks whether the solve goes on before it pays for the assembly of the
  // reduced matrix at the operator switch; nothing of the iteration that follows is started (the prologue that starts
  // it evaluates the same sums again and comes to the same verdict).
  const int lane = threadIdx.x & 63;
  const double prho = part_rho[lane], pq1 = part_q[lane];
  const int done = st->done, it = st->iter, need_test = st->need_test;
  const double q_prev = st->q_hist[(it + 1) & 1], rho_prev = st->rho_hist[(it + 1) & 1];
  const double rho = wave_sum(prho), q1 = wave_sum(pq1);
  int term = 0, res_it = it, own_stop = 0;
  double beta = 0.0;
  if (!done) {
    if (need_test) {
      // Q-model test (conjugate_gradient.hpp:239-276); residual-based test is off (r_tolerance = -1)
      const double zeta = it * (q1 - q_prev) / q1;
      // (the host's early operator switch looks at the trend of zeta over iterations 3 and 4, Solver::pcg)
      if (host_progress && (it == 3 || it == 4) && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(host_progress + it - 1, __float_as_int(float(zeta)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (zeta < q_tolerance && it >= min_it) {
        own_stop = 1;
        term = 1;
      } else if (it >= max_it) {
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
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (done) {
      if (host_progress) __hip_atomic_store(host_progress + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      if (need_test) st->q_hist[it & 1] = q1;
      if (own_stop) {
        st->termination = term;
        st->result_iter = res_it;
        st->done = 1;
        if (host_progress) __hip_atomic_store(host_progress + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        if (!test_only) {
          st->rho_hist[it & 1] = rho;
          st->beta = beta;
          st->cur = it + 1;
        }
        // (RELEASE: the zeta words above are visible to a host that has seen this progress value - it reads them behind an
        //  acquire fence, Solver::pcg; all ranks must take the same early-switch decision)
        if (host_progress) __hip_atomic_store(host_progress, it + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  if (done | own_stop | test_only) return;
  const S bs = S(beta);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const S pn = it == 0 ? z[i] : z[i] + bs * pvec[i];
    pvec[i] = pn;
    if (pscaled) pscaled[i] = dscale[i] * pn;  // compact stage 2: the operand of the matrix-free product is D p
    q[i] = S(0);
  }
}

// Start of a solve in one kernel (round 1: k_pcg_init, k_pcgs_begin, k_pcg_a1): x = 0, r = b, z = M^-1 b with the
// partials of rho = r.z (28 cameras per 252-thread tile, as in k_pcgs_update), and the reset state - |b|^2 is summed by
// every workgroup alike, workgroup 0 writes the state.
template <class S>
__global__ __launch_bounds__(256) void k_pcgs_start(const S* __restrict__ inv, const S* __restrict__ bvec,
                                                    S* __restrict__ x, S* __restrict__ r, S* __restrict__ z, int n_cams,
                                                    CgState* st, double* __restrict__ part_rho, double lambda,
                                                    int pswap, unsigned long long* __restrict__ stamp) {
  stage_stamp(stamp);
  __shared__ double sm[4];
  __shared__ S rl[252];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = 9 * n_cams, n_tiles = (n_cams + 27) / 28;
  double accb = 0;
  if (blockIdx.x == 0) {
    for (int base = 0; base < n; base += 2048) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256 + tid;
        const S bi = bvec[min(idx, n - 1)];
        v[u] = double(bi) * double(bi) * (idx < n ? 1.0 : 0.0);
      }
      accb += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
  }
  double acc_rho = 0;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int i = 252 * tile + tid;
    const bool act = tid < 252 && i < n;
    const int c = act ? i / 9 : 0, row = act ? i - 9 * c : 0;
    S bi = S(0), Mrow[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) Mrow[j] = S(0);
    if (act) {
      bi = bvec[i];
#pragma unroll
      for (int j = 0; j < 9; ++j) Mrow[j] = inv[81 * c + 9 * row + j];
      x[i] = S(0);
      r[i] = bi;
    }
    __syncthreads();
    if (tid < 252) rl[tid] = bi;
    __syncthreads();
    if (act) {
      const S* rc = rl + 9 * (tid / 9);
      S zc = S(0);
#pragma unroll
      for (int j = 0; j < 9; ++j) zc += Mrow[j] * rc[j];
      z[i] = zc;
      acc_rho += double(bi) * double(zc);
    }
  }
  const double t0 = wave_sum(acc_rho), t1 = wave_sum(accb);
  __syncthreads();
  if (lane == 0) sm[wave] = t0;
  __syncthreads();
  if (tid == 0) part_rho[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  if (blockIdx.x != 0) return;
  __syncthreads();
  if (lane == 0) sm[wave] = t1;
  __syncthreads();
  if (tid == 0) {
    const double nb2 = (sm[0] + sm[1]) + (sm[2] + sm[3]);
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
    st->lambda = lambda;
    st->pswap = pswap;
  }
}

// per-solve parameters (see CgState)
__global__ void k_pcgs_begin(CgState* st, double lambda, int pswap) {
  st->lambda = lambda;
  st->pswap = pswap;
}

// y = q (+ the extra items of long rows)   (tests: rba_right_multiply_explicit)
template <class S>
__global__ __launch_bounds__(256) void k_pcgs_collect(S* __restrict__ y, QPieces<S> qp, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = i / 9, row = i - 9 * c;
  y[i] = pcgs_gather_q(qp, c, row);
}

// r = b - (S x + lambda x) from a refresh product (operator switch inside a solve)
template <class S>
__global__ __launch_bounds__(256) void k_pcgs_residual(const S* __restrict__ bvec, S* __restrict__ r, QPieces<S> qp,
                                                       int n, const CgState* st) {
  if (st->done) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = i / 9, row = i - 9 * c;
  r[i] = bvec[i] - pcgs_gather_q(qp, c, row);
}

// One term of the power-series preconditioner z = sum_{i=0..m} (Hpp^-1 E0)^i Hpp^-1 r (PowerSCPreconditioner::solve_assign,
// src/rootba/cg/preconditioner.hpp:180-192) through the assembled matrix: with E0 = Hpp + lambda I - (S + lambda I),
//   t <- (Hpp^-1 E0) t = t - Hpp^-1 w,  w = (S + lambda I) t  (the plain product k_pcgs_spmv<2>, pieces gathered here);  z += t.
// Tiles of 28 cameras like k_pcgs_update (the 9 entries of a camera's w are exchanged through LDS). The LAST term also
// leaves the partials of rho = r.z for the next direction update.
template <class S>
__global__ __launch_bounds__(256) void k_pcgs_series_step(const S* __restrict__ inv, QPieces<S> qp, S* __restrict__ t,
                                                          S* __restrict__ z, const S* __restrict__ r, int n_cams,
                                                          const CgState* st, int last, double* __restrict__ part_rho,
                                                          int direct = 0) {
  // `direct`: the product is E0 t itself (matrix-free term, k_e0*: one complete vector in qp.qmain), the next term
  // Hpp^-1 (E0 t); otherwise it is (S + lambda I) t from the assembled matrix and the next term t - Hpp^-1 of it
  __shared__ double sm[4];
  __shared__ S wl[252];
  if (st->done) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = 9 * n_cams, n_tiles = (n_cams + 27) / 28;
  double acc_rho = 0;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int i = 252 * tile + tid;
    const bool act = tid < 252 && i < n;
    const int c = act ? i / 9 : 0, row = act ? i - 9 * c : 0;
    S Mrow[9], ti = S(0), zi = S(0), ri = S(0), wi = S(0);
#pragma unroll
    for (int j = 0; j < 9; ++j) Mrow[j] = S(0);
    if (act) {
#pragma unroll
      for (int j = 0; j < 9; ++j) Mrow[j] = inv[81 * c + 9 * row + j];
      ti = t[i];
      zi = z[i];
      if (last) ri = r[i];
      wi = direct ? qp.qmain[i] : pcgs_gather_q(qp, c, row);
    }
    __syncthreads();
    if (tid < 252) wl[tid] = wi;
    __syncthreads();
    if (act) {
      const S* wc = wl + 9 * (tid / 9);
      S v = S(0);
#pragma unroll
      for (int j = 0; j < 9; ++j) v += Mrow[j] * wc[j];
      const S tn = direct ? v : ti - v, zn = zi + tn;
      t[i] = tn;
      z[i] = zn;
      acc_rho += double(ri) * double(zn);
    }
  }
  if (!last) return;
  const double t0 = wave_sum(acc_rho);
  __syncthreads();
  if (lane == 0) sm[wave] = t0;
  __syncthreads();
  if (tid == 0) part_rho[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// Half storage, heavy rows (see the top of the file): one wavefront sums the received slots of a row in a fixed order
// (lane-strided partial sums, then the DPP tree) into the row's additional "further item". Same early-out conditions as
// the product it follows.
template <class S>
__global__ __launch_bounds__(256) void k_pcgs_reduce_slots(const HeavyRow* __restrict__ rows, int n_rows,
                                                           const double* __restrict__ tpart, S* __restrict__ qextra,
                                                           const CgState* st, int mode, int period) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int h = blockIdx.x * 4 + wave;
  if (h >= n_rows) return;
  if (st->done || (mode == 1 && st->cur % period != 0)) return;
  const HeavyRow r = rows[h];
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int s = r.slot0 + lane; s < r.slot1; s += 64) {
    const double* t = tpart + size_t(9) * s;
#pragma unroll
    for (int a = 0; a < 9; ++a) acc[a] += t[a];
  }
#pragma unroll
  for (int a = 0; a < 9; ++a) {
    const double tot = wave_sum(acc[a]);
    if (lane == a) qextra[9 * r.extra + a] = S(tot);
  }
}

}  // namespace rba
