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
//   * there is NO grid barrier. Everything exchanged travels as 16-byte self-validating RECORDS - {double, tag, check} or
//     {three floats, tag ^ check} - written by ONE write-through store (sc1) and read by L1-bypassing loads until tag and
//     check are this iteration's (the data is the flag; a record torn into its 8-byte halves fails the check). The
//     workgroups' partial sums are published in EIGHT replicas (one store instruction) and a workgroup polls replica
//     (g mod 8): 230 workgroups hammering the same 58 cache lines was the largest cost of a first version
//     (scripts/microbench/allgather.hip: 3.3 us per all-gather with 8-byte granules, 2.3 us with 16-byte records,
//     1.75 us with replicas; scripts/microbench/grid_barrier.hip: 7.1 us for the cheapest counter barrier pair, 14-17 us
//     with release / acquire fences). Overwriting is safe without double buffering: the two all-gathers of an iteration
//     guard each other - nobody publishes exchange k + 1 before it has read EVERY workgroup's record of exchange k,
//     which those workgroups wrote after reading exchange k - 1;
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
#include "pg_record_io.hpp"

namespace rba {


constexpr int kPgThreads = 512;               // 8 wavefronts: two per SIMD, one workgroup per CU
constexpr int kPgMaxRows = kPgThreads / 9;    // nine row outputs per camera
constexpr int kPgQuads = kPgThreads / 4;
constexpr int kPgMaxGroups = 256;             // workgroups (a record per lane of four polling wavefronts)
constexpr int kPgReplicas = 8;                // copies of every workgroup's partial sums
constexpr pg_u32 kPgSpinLimit = 1u << 20;     // sweeps of ~1 us before a workgroup gives up
constexpr int kPgTraceIts = 64;

struct PgWorkgroup {
  int row0, nrows;  // cameras row0 .. row0 + nrows - 1
  int ncols;        // distinct columns its blocks multiply (staged in LDS by work-items 0 .. ncols - 1)
  int pad;
};

// records per camera of an exchanged vector: three floats or one double per record
template <class S>
constexpr int pg_vec_records() {
  return sizeof(S) == 4 ? 3 : 9;
}

template <class S>
struct PgParams {
  const PgWorkgroup* wg;           // [G]
  const int* lane_src;             // [G][512]  2 * slot + transposed of the lane's block in the half-storage matrix; -1: padding
  const unsigned short* lane_col;  // [G][512]  staged column (index into the workgroup's list) of the lane's block
  const int* stage_col;            // [G][512]  camera of staged column t; -1 beyond ncols
  const int* row_info;             // [n_c][3]  first quad of the row in its workgroup, quads, staged index of its own column
  const void* vals;                // [nnz][81]: double (half storage of the square-root solver) or the solver's scalar
  int vals_solver_scalar;          //            (full storage of the explicit-SC backend)
  const S* inv;                    // M^-1 [n_c][81]
  const S* b;
  S* x;                            // in: iterate after `iter` iterations; out: the solution
  const S* r_in;                   // residual (not read when the operator is switched: recomputed)
  const S* p_in;                   // direction of the last completed iteration
  pg_rec* zg;                      // [n_c][3 | 9] records of z
  pg_rec* xg;                      // [n_c][3 | 9] records of x (refresh product)
  pg_rec* tg;                      // [2][n_c][3 | 9] records of the power-series term t, two buffers by the parity of the term
  pg_rec* part_rq;                 // [8][G][2]  partial sums of rho and Q, eight replicas
  pg_rec* part_pq;                 // [8][G]     partial sums of p.q
  CgState* st;
  int* host_progress;              // pinned: [1] done, [4] aborted
  pg_u32 tag_base;
  int tag_stride;                  // tags per iteration: 1 + series + 1 (tag_base + iteration * stride: z, sums, x; + i: term i)
  int series;                      // terms of the power-series preconditioner behind z = M^-1 r (0: block-diagonal M^-1 alone)
  int G, n_cams;
  int switch_operator;             // the solve ran matrix-free so far: r = b - (S + lambda I) x first (like the refresh)
  double q_tolerance;
  int min_it, max_it, period;
  long long* trace;                // debug (RBA_PCGP_TRACE): 100 MHz real-time stamps of the first kPgTraceIts iterations, 8 per
                                   // iteration and workgroup; nullptr in production
};

// {double, tag, check}: check = tag ^ hi ^ lo, so a record torn into its halves does not pass
__device__ __forceinline__ pg_rec pg_pack(double v, pg_u32 tag) {
  const pg_u64 bits = pg_u64(__double_as_longlong(v));
  const pg_u32 lo = pg_u32(bits), hi = pg_u32(bits >> 32);
  pg_rec r;
  r.x = lo;
  r.y = hi;
  r.z = tag;
  r.w = tag ^ hi ^ lo;
  return r;
}
__device__ __forceinline__ bool pg_unpack(pg_rec r, pg_u32 tag, double& v) {
  v = __longlong_as_double((long long)((pg_u64(r.y) << 32) | pg_u64(r.x)));
  return r.z == tag && r.w == (tag ^ r.y ^ r.x);
}
// {f0, f1, f2, tag ^ f0 ^ f1 ^ f2}: old and new halves mixed pass only if the mixed-in words did not change (or by a
// 2^-32 coincidence)
__device__ __forceinline__ pg_rec pg_pack3(float a, float b, float c, pg_u32 tag) {
  pg_rec r;
  r.x = __float_as_uint(a);
  r.y = __float_as_uint(b);
  r.z = __float_as_uint(c);
  r.w = tag ^ r.x ^ r.y ^ r.z;
  return r;
}
__device__ __forceinline__ bool pg_unpack3(pg_rec r, pg_u32 tag, float& a, float& b, float& c) {
  a = __uint_as_float(r.x);
  b = __uint_as_float(r.y);
  c = __uint_as_float(r.z);
  return (r.w ^ r.x ^ r.y ^ r.z) == tag;
}
// the nine entries of a camera <-> its records
__device__ __forceinline__ bool pg_unpack_vec(const pg_rec (&r)[3], pg_u32 tag, float (&v)[9]) {
  bool ok = pg_unpack3(r[0], tag, v[0], v[1], v[2]);
  ok &= pg_unpack3(r[1], tag, v[3], v[4], v[5]);
  ok &= pg_unpack3(r[2], tag, v[6], v[7], v[8]);
  return ok;
}
__device__ __forceinline__ bool pg_unpack_vec(const pg_rec (&r)[9], pg_u32 tag, double (&v)[9]) {
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 9; ++a) ok &= pg_unpack(r[a], tag, v[a]);
  return ok;
}

// Block rows kept in LDS instead of registers (float solver: two of the nine - 36 registers that the row wave's step needs
// to hold the row's state across the gather of p.q; the double solver's vectors leave no LDS for them)
template <class S>
constexpr int pg_lds_rows() {
  return sizeof(S) == 4 ? 2 : 0;
}
template <class S>
constexpr size_t pgp_lds_bytes() {
  return size_t(pg_lds_rows<S>()) * 9 * kPgThreads * sizeof(double)  // blkl
         + size_t(3) * kPgThreads * 9 * sizeof(S)   // pst, zst, minv
         + size_t(6) * kPgThreads * sizeof(S)     // xs, rs, bs, pcs, qss, ts
         + size_t(kPgQuads) * 9 * sizeof(double)  // red
         + 24 * sizeof(double)                    // bc, gs, wpq
         + size_t(3) * kPgMaxRows * sizeof(int)   // rowtab
         + 8 * sizeof(int)                        // endi
         + 16;                                    // flags
}

// p = z + beta p_old as ONE fused multiply-add - the block lanes, the row sums and the stagers all form it, bit for bit
__device__ __forceinline__ float pg_dir(float z, float b, float p) { return __builtin_fmaf(b, p, z); }
__device__ __forceinline__ double pg_dir(double z, double b, double p) { return __builtin_fma(b, p, z); }

// sum over the four lanes of a quad, every lane receives it (fixed order: (l0 + l1) + (l2 + l3) up to commutation)
// (a quad permutation writes every lane: no initial value for the destination - dpp_mov0 spends a v_mov on it per word)
template <int CTRL>
__device__ __forceinline__ double pg_dpp_perm(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_mov_dpp(int(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(int(b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}
__device__ __forceinline__ double pg_quad_sum(double v) {
  v += pg_dpp_perm<0xb1>(v);  // quad_perm:[1,0,3,2]
  v += pg_dpp_perm<0x4e>(v);  // quad_perm:[2,3,0,1]
  return v;
}
__device__ __forceinline__ int pg_flag(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Roles inside a workgroup (three workgroup barriers per iteration: exchange 1 done / quad sums written / row sums written):
//   every lane        one block of the matrix: forms the direction p_j = z_j + beta p_j of ITS column on the fly from the
//                     staged z and the previous direction (LDS), multiplies, quad sums into LDS
//   work-items 0 ..   "stagers": one staged column each - poll its z records into LDS, keep its direction p in LDS
//   waves 4 - 7       gather the workgroups' partial sums of rho and Q (a workgroup per lane); everybody then takes the
//                     decisions from the four wave sums
//   four per output   the row sums of the product: output j = 9 row + a is summed by four work-items over the row's quads
//                     (interleaved, then the quad's DPP sum), with p_c q_c for the dot product
//   the last wave     "row wave": x, r, b, M^-1 of the workgroup's rows in LDS; p.q with its gather, the step, z = M^-1 r,
//                     the workgroup's partial sums. One wavefront, two outputs per lane and pass: no workgroup barrier.
template <class S>
__global__ __launch_bounds__(kPgThreads) void k_pcgp(PgParams<S> P) {
  constexpr int NR = pg_vec_records<S>();
  extern __shared__ __attribute__((aligned(16))) char smem_pg[];
  constexpr int LR = pg_lds_rows<S>(), RR = 9 - LR;  // block rows in LDS / in registers
  double* blkl = reinterpret_cast<double*>(smem_pg);  // [LR * 9][512] the last LR rows of every lane's block, entry-major
  S* pst = reinterpret_cast<S*>(blkl + LR * 9 * kPgThreads);  // [512][9] direction p of the staged columns (kept across iterations)
  S* zst = pst + kPgThreads * 9;            // [512][9] z of the staged columns; x during a refresh product
  S* minv = zst + kPgThreads * 9;           // [512][9] row a of M^-1 of row output j = 9 row + a
  S* xs = minv + kPgThreads * 9;            // [512]    x, r, b of the row outputs; p_c and q_c of the current iteration
  S* rs = xs + kPgThreads;
  S* bs = rs + kPgThreads;
  S* pcs = bs + kPgThreads;                 //          (also: z of the row outputs on its way into the records)
  S* qss = pcs + kPgThreads;
  S* ts = qss + kPgThreads;                 //          the current term of the power series
  double* red = reinterpret_cast<double*>(ts + kPgThreads);  // [128][9] quad sums of the block products
  double* bc = red + kPgQuads * 9;                            // [3] rho_prev [4] q_prev; at the end [0] beta [1] rho [2] q1 [5] p.q [6] alpha
  double* gs = bc + 8;                                        // [4][2] sums of rho and Q partial sums of the four polling waves
  double* wpq = gs + 8;                                       // [8] p.q of the row outputs summed by the waves that formed them
  int* rowtab = reinterpret_cast<int*>(wpq + 8);              // [56][3] first quad, quads, staged index of the own column
  int* endi = rowtab + 3 * kPgMaxRows;                        // end of the solve: [0] termination [1] result_iter [2] indefinite [3] stepped
  int* sflag = endi + 8;                                      // [0] abort, [1] the row wave has ended the solve

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = blockIdx.x;
  const PgWorkgroup W = P.wg[g];
  const int src = P.lane_src[size_t(g) * kPgThreads + tid];
  const bool act = src >= 0;
  const int scol = P.lane_col[size_t(g) * kPgThreads + tid];
  const int mycol = P.stage_col[size_t(g) * kPgThreads + tid];
  const bool stager = mycol >= 0;
  const bool wave_stages = wave * 64 < W.ncols;  // (the staged columns are work-items 0 .. ncols - 1)
  const bool roww = wave == kPgThreads / 64 - 1;
  const int nout = 9 * W.nrows;
  const int G = P.G;
  CgState* st = P.st;
  if (st->done) return;  // (uniform over the grid: nobody writes the state before the end)
  if (tid < 2) sflag[tid] = 0;

  // ---- the lane's block: 81 doubles for the whole solve - rows 0 .. RR - 1 in registers, the rest in LDS -----------------
  double blk[9 * RR];
  {
    const size_t v0 = size_t(81) * size_t(act ? (src >> 1) : 0);
    const double* vd = static_cast<const double*>(P.vals) + v0;
    const S* vs = static_cast<const S*>(P.vals) + v0;
    const bool tr = (src & 1) != 0, vss = P.vals_solver_scalar != 0;
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
      for (int bb = 0; bb < 9; ++bb) {
        const int ei = tr ? 9 * bb + a : 9 * a + bb;
        const double t = vss ? double(vs[ei]) : vd[ei];
        const double e = act ? t : 0.0;
        if (a < RR)
          blk[9 * a + bb] = e;
        else
          blkl[(9 * (a - RR) + bb) * kPgThreads + tid] = e;
      }
  }
  // ---- state -----------------------------------------------------------------------------------------------------
  int it = st->iter, need_test = st->need_test;
  if (tid == 0) {
    bc[3] = st->rho_hist[(it + 1) & 1];
    bc[4] = st->q_hist[(it + 1) & 1];
  }
  const S lambda = S(st->lambda);
  if (roww) {
    for (int j = lane; j < nout; j += 64) {
      const int row = j / 9, a = j - 9 * row, c = W.row0 + row;
      xs[j] = P.x[9 * c + a];
      bs[j] = P.b[9 * c + a];
      rs[j] = P.switch_operator ? S(0) : P.r_in[9 * c + a];
#pragma unroll
      for (int b2 = 0; b2 < 9; ++b2) minv[9 * j + b2] = P.inv[81 * c + 9 * a + b2];
    }
    for (int row = lane; row < W.nrows; row += 64) {
      rowtab[3 * row] = P.row_info[3 * (W.row0 + row)];
      rowtab[3 * row + 1] = P.row_info[3 * (W.row0 + row) + 1];
      rowtab[3 * row + 2] = P.row_info[3 * (W.row0 + row) + 2];
    }
  }
  if (stager) {
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      pst[9 * tid + a] = it > 0 ? P.p_in[9 * mycol + a] : S(0);
      zst[9 * tid + a] = P.switch_operator ? P.x[9 * mycol + a] : S(0);
    }
  }

  int rcon = 0;
  // the lane's share of q = S v: quad sums into `red` (the caller puts the barrier). v_j = z_j + bsel p_j formed here from the
  // staged vectors: bsel = beta for a direction product (0 in the first iteration, whose staged p is 0), 0 for the
  // refresh product of x. No branch on that: a conditional around an LDS read is a round trip of its own - nine of them
  // in a first version (0.96 us for a product whose arithmetic takes 0.3).
  auto product = [&](S bsel) {
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (act) {
      // (three columns at a time: six LDS reads in flight, nine accumulators and three operand entries live beside the
      //  162 registers of the block)
#pragma unroll
      for (int b0 = 0; b0 < 9; b0 += 3) {
        S zz[3], pp[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          zz[u] = zst[9 * scol + b0 + u];
          pp[u] = pst[9 * scol + b0 + u];
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const double pv = double(pg_dir(zz[u], bsel, pp[u]));
#pragma unroll
          for (int a = 0; a < RR; ++a) acc[a] += blk[9 * a + b0 + u] * pv;
#pragma unroll
          for (int a = RR; a < 9; ++a) acc[a] += blkl[(9 * (a - RR) + b0 + u) * kPgThreads + tid] * pv;
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 9; ++a) acc[a] = pg_quad_sum(acc[a]);
    // every lane of the quad holds the sums: lane q writes entries q, q + 4, q + 8
    double* dst = red + 9 * (tid >> 2);
    const int ql = lane & 3;
    dst[ql] = ql == 0 ? acc[0] : ql == 1 ? acc[1] : ql == 2 ? acc[2] : acc[3];
    dst[ql + 4] = ql == 0 ? acc[4] : ql == 1 ? acc[5] : ql == 2 ? acc[6] : acc[7];
    if (ql == 0) dst[8] = acc[8];
  };
  // the row sums of a product, by ALL work-items: four per output over the row's quads (interleaved), the quad's DPP sum
  // closes them (fixed order). DIRECTION: q_c = sum + lambda p_c with p_c as the product formed it, and p_c q_c for the
  // dot product, summed per wave (wpq); otherwise (refresh) q_c = sum + lambda x_c. The caller puts the barrier.
  // (`rcon`: first quad | quads << 8 | staged own column << 16 | a << 28 of the work-item's output in the first chunk -
  //  the table lookup would be an LDS round trip ahead of the sums)
  auto rowsums = [&](int mode, S bsel) {  // 0: direction, 1: refresh (operand x), 2: term of the power series (operand t)
    const bool direction = mode == 0;
    const int part = tid & 3;
    double my_pq = 0.0;
    for (int cb = 0; cb < nout; cb += kPgThreads / 4) {
      const int j = cb + (tid >> 2);
      const bool on = j < nout;
      const int jj = on ? j : 0;
      int a, q0, nq, self;
      if (cb == 0) {
        q0 = rcon & 0xff;
        nq = (rcon >> 8) & 0xff;
        self = (rcon >> 16) & 0xfff;
        a = (rcon >> 28) & 0xf;
      } else {
        const int row = jj / 9;
        a = jj - 9 * row;
        q0 = rowtab[3 * row];
        nq = rowtab[3 * row + 1];
        self = rowtab[3 * row + 2];
      }
      const S zc = zst[9 * self + a], po = pst[9 * self + a], xc = mode == 2 ? ts[jj] : xs[jj];  // (all loads ahead of the sums)
      double q = 0.0;
      for (int k0 = part; k0 < nq; k0 += 24) {  // (six loads in flight: ONE pass for rows of up to 96 blocks)
        double v[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) v[u] = red[9 * (q0 + min(k0 + 4 * u, nq - 1)) + a] * (k0 + 4 * u < nq ? 1.0 : 0.0);
        q += ((v[0] + v[1]) + (v[2] + v[3])) + (v[4] + v[5]);
      }
      q = pg_quad_sum(q);
      if (on && part == 0) {
        const S pc = direction ? pg_dir(zc, bsel, po) : xc;
        S qs = S(q);
        qs += lambda * pc;  // pose damping term of right_multiply
        qss[j] = qs;
        if (direction) {
          pcs[j] = pc;
          my_pq += double(pc) * double(qs);
        }
      }
    }
    if (direction) {
      const double t = wave_sum(my_pq);
      if (lane == 0) wpq[wave] = t;
    }
  };
  // row wave: the entries of a vector of the workgroup's rows (LDS, [9 nrows]) as records, published with `tagn`
  auto publish_vec = [&](pg_rec* dst, const S* v, pg_u32 tagn) {
    wave_lds_fence();  // (v of the other lanes)
    for (int t = lane; t < NR * W.nrows; t += 64) {
      const int row = t / NR, k = t - NR * row;
      pg_rec r;
      if constexpr (NR == 3)
        r = pg_pack3(float(v[9 * row + 3 * k]), float(v[9 * row + 3 * k + 1]), float(v[9 * row + 3 * k + 2]), tagn);
      else
        r = pg_pack(double(v[9 * row + k]), tagn);
      pg_rec_store(dst + size_t(NR) * size_t(W.row0 + row) + k, r);
    }
  };
  // row wave: z = M^-1 r (the power series: its first term t as well)
  auto close_first_term = [&]() {
    wave_lds_fence();  // (rs of the other lanes)
    for (int j0 = lane; j0 < nout; j0 += 128) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {  // (two outputs per pass: their LDS round trips overlap)
        const int j = j0 + 64 * u;
        if (j < nout) {
          const S* rc = rs + 9 * (j / 9);
          S zc = S(0);
#pragma unroll
          for (int b2 = 0; b2 < 9; ++b2) zc += minv[9 * j + b2] * rc[b2];
          pcs[j] = zc;
          ts[j] = zc;
        }
      }
    }
  };
  // row wave: the partial sums of rho = r.z and Q = -x.(b + r) and z itself, published for the iteration with tag `tagn`
  auto close_publish = [&](pg_u32 tagn) {
    double acc_rho = 0.0, acc_q = 0.0;
    for (int j = lane; j < nout; j += 64) {
      const S r_i = rs[j], x_i = xs[j];
      acc_rho += double(r_i) * double(pcs[j]);
      acc_q -= double(x_i) * double(bs[j] + r_i);
    }
    publish_vec(P.zg, pcs, tagn);
    const double s0 = wave_sum(acc_rho), s1 = wave_sum(acc_q);
    if (lane < 2 * kPgReplicas) {
      const int rep = lane >> 1, k = lane & 1;
      pg_rec_store(P.part_rq + (size_t(rep) * G + g) * 2 + k, pg_pack(k ? s1 : s0, tagn));
    }
  };
  // what a wavefront does when its sweep did not find this exchange's tags: wait a little; 0 = sweep again, 1 = the row
  // wave has ended the solve (nothing more will come), 2 = give up (after ~1 s, or somebody else did). Wave-uniform.
  auto poll_again = [&](pg_u32& spins) -> int {
    __builtin_amdgcn_s_sleep(1);
    ++spins;
    int code = pg_flag(sflag + 1) != 0 ? 1 : 0;
    if (pg_flag(sflag) != 0 || spins > kPgSpinLimit) code = 2;
    if ((spins & 1023u) == 0 && __hip_atomic_load(P.host_progress + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) code = 2;
    return __ballot(code == 2) != 0 ? 2 : __ballot(code == 1) != 0 ? 1 : 0;
  };
  auto raise_abort = [&]() {
    if (lane == 0) {
      __hip_atomic_store(P.host_progress + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(sflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  // stagers: the records of the staged column from `src` into zst (wave-uniform call)
  auto stage_vector = [&](const pg_rec* srcv, pg_u32 tag) {
    pg_u32 spins = 0;
    S zv[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) zv[a] = S(0);
    for (;;) {
      bool ok = true;
      if (stager) {
        unsigned mc = unsigned(mycol);
        PG_OPAQUE(mc);
        pg_rec r[NR];
        pg_rec_load(srcv + size_t(NR) * mc, 1, r);
        ok = pg_unpack_vec(r, tag, zv);
      }
      if (__ballot(!ok) == 0) break;
      const int pa = poll_again(spins);  // (1: the row wave ended the solve at the step length - nothing will come)
      if (pa == 2) raise_abort();
      if (pa) break;
    }
    if (stager) {
#pragma unroll
      for (int a = 0; a < 9; ++a) zst[9 * tid + a] = zv[a];
    }
  };
  // the end of the solve: the final state is parked in LDS by whoever decides, written out by the row wave
  int my_stop = 0;
  auto finish = [&]() {  // row wave
    for (int j = lane; j < nout; j += 64) {
      const int row = j / 9, a = j - 9 * row;
      P.x[9 * (W.row0 + row) + a] = xs[j];
    }
    if (g == 0 && lane == 0) {
      if (need_test) st->q_hist[it & 1] = bc[2];
      if (endi[3]) {  // iteration it + 1 was started (direction and product done) when the step came out unusable
        st->rho_hist[it & 1] = bc[1];
        st->beta = bc[0];
        st->cur = it + 1;
        st->pq = bc[5];
        st->alpha = bc[6];
      }
      st->iter = it;
      st->need_test = need_test;
      st->termination = endi[0];
      st->indefinite = endi[2];
      st->result_iter = endi[1];
      st->done = 1;
      __hip_atomic_store(P.host_progress + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  };
  const int it_first = it;
  auto stamp = [&](int k) {
    if (P.trace && tid == kPgThreads - 64 && it - it_first < kPgTraceIts)
      P.trace[(size_t(g) * kPgTraceIts + size_t(it - it_first)) * 8 + k] =
          wall_clock64();  // (s_memrealtime: 100 MHz, ONE counter for the chip - stamps of different workgroups compare)
  };

  // EVERYBODY: z = M^-1 r of the workgroup's rows for the iteration with tag `tagn`. With the power-series preconditioner
  // (PowerSCPreconditioner::solve_assign, src/rootba/cg/preconditioner.hpp:180-192, through the matrix as in
  // k_pcgs_series_step): z = sum_i t_i, t_0 = Hpp^-1 r, t_i = t_i-1 - Hpp^-1 (S + lambda I) t_i-1 - a product per term,
  // whose operand travels like z in one of TWO record buffers (parity of the term: a workgroup that writes term i + 2
  // has read term i + 1 of all its neighbours, who wrote it after reading its term i - the matrix is structurally
  // symmetric, so whoever reads my rows is read by me). Returns false when the kernel has to end.
  auto close_all = [&](pg_u32 tagn) -> bool {
    const bool live = roww && !my_stop;
    if (live) close_first_term();
    for (int i = 1; i <= P.series; ++i) {
      const pg_u32 ttag = tagn - pg_u32(P.tag_stride) + pg_u32(i);
      pg_rec* tb = P.tg + ((i & 1) ? size_t(0) : size_t(NR) * size_t(P.n_cams));
      if (live) publish_vec(tb, ts, ttag);
      if (wave_stages) stage_vector(tb, ttag);
      __syncthreads();
      if (pg_flag(sflag) != 0) return false;
      if (pg_flag(sflag + 1) != 0) {
        if (roww) finish();
        return false;
      }
      product(S(0));
      __syncthreads();
      rowsums(2, S(0));
      __syncthreads();
      if (live) {
        for (int j = lane; j < nout; j += 64) {
          const S* wc = qss + 9 * (j / 9);
          S v = S(0);
#pragma unroll
          for (int b2 = 0; b2 < 9; ++b2) v += minv[9 * j + b2] * wc[b2];
          const S tn = ts[j] - v;
          ts[j] = tn;
          pcs[j] += tn;
        }
      }
    }
    if (live) close_publish(tagn);
    return true;
  };

  __syncthreads();  // staged vectors / row state / flags
  {
    const int j = tid >> 2, jj = j < nout ? j : 0, row = jj / 9;
    rcon = rowtab[3 * row] | (rowtab[3 * row + 1] << 8) | (rowtab[3 * row + 2] << 16) | ((jj - 9 * row) << 28);
  }
  if (P.switch_operator) {
    // operator switch inside a running solve: the residual is recomputed with the operator used from here on,
    // r = b - (S + lambda I) x, exactly like the periodic refresh (Solver::pcg_fused)
    product(S(0));
    __syncthreads();
    rowsums(1, S(0));
    __syncthreads();
    if (roww) {
      unsigned j0 = unsigned(lane);
      PG_OPAQUE(j0);  // (its own copy of the lane index, made here: the one kept for these two cold loops was the kernel's last spill)
      for (int j = int(j0); j < nout; j += 64) rs[j] = bs[j] - qss[j];
    }
  }
  if (!close_all(P.tag_base + pg_u32(it + 1) * pg_u32(P.tag_stride))) return;

  for (;;) {
    const int cur = it + 1;
    const pg_u32 tag = P.tag_base + pg_u32(cur) * pg_u32(P.tag_stride);
    const bool refresh = (cur % P.period) == 0;
    stamp(0);
    // ---- exchange 1: z of the staged columns (stagers); partial sums of rho and Q and the decisions (row wave) ----------
    if (!my_stop) {
      if (roww) {
        pg_u32 spins = 0;
        double a_rho = 0.0, a_q = 0.0;
        for (;;) {
          unsigned l0 = unsigned(lane);
          PG_OPAQUE(l0);
          const pg_rec* base = P.part_rq + size_t(g % kPgReplicas) * G * 2;
          const int last = G - 1;  // (clamped, masked below: lanes beyond G re-read the last pair)
          pg_rec r[8];
          pg_rec_load_pairs(base + 2 * min(int(l0), last), base + 2 * min(int(l0) + 64, last), base + 2 * min(int(l0) + 128, last),
                            base + 2 * min(int(l0) + 192, last), r);
          bool ok = true;
          double v[8];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bool on = int(l0) + 64 * k < G;
            const bool good0 = pg_unpack(r[2 * k], tag, v[2 * k]), good1 = pg_unpack(r[2 * k + 1], tag, v[2 * k + 1]);
            ok &= (good0 && good1) || !on;
            if (!on) v[2 * k] = v[2 * k + 1] = 0.0;
          }
          a_rho = (v[0] + v[2]) + (v[4] + v[6]);
          a_q = (v[1] + v[3]) + (v[5] + v[7]);
          if (__ballot(!ok) == 0) break;
          const int pa = poll_again(spins);
          if (pa == 2) raise_abort();
          if (pa) break;
        }
        // ---- decisions (k_pcgs_spmv<0> prologue): test of the previous iteration, rho, beta - by ONE wave, ahead of
        //      the barrier (by everybody behind it they were 0.6 us of every iteration: eight waves, two per SIMD, each
        //      through a double-precision division and a dozen branches) ---------------------------------------------------
        const double rho_prev = bc[3], q_prev = bc[4];  // (read by every lane BEFORE lane 0 replaces them below)
        const double rho = wave_sum(a_rho), q1 = wave_sum(a_q);
        double beta = 0.0;
        int own_stop = 0, term = 0, res_it = it;
        if (need_test) {
          // Q-model test (conjugate_gradient.hpp:239-276); residual-based test is off (r_tolerance = -1)
          // zeta = it (q1 - q_prev) / q1 < tolerance without the division: multiplied through by q1, the inequality
          // turned for q1 < 0; q1 = 0 (x = 0) never passes
          const double num = it * (q1 - q_prev), bound = P.q_tolerance * q1;
          const bool small = q1 > 0.0 ? num < bound : (q1 < 0.0 && num > bound);
          if (small && it >= P.min_it) {
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
        if (lane == 0) {
          bc[0] = beta;
          bc[1] = rho;
          bc[2] = q1;
          if (!own_stop) {
            bc[3] = rho;
            if (need_test) bc[4] = q1;
          } else {
            endi[0] = term;
            endi[1] = res_it;
            endi[2] = 0;
            endi[3] = 0;
            __hip_atomic_store(sflag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        if (own_stop) my_stop = 1;
      }
      if (wave_stages) stage_vector(P.zg, tag);
    }
    stamp(1);
    __syncthreads();  // #1: z is staged, beta is known
    // (ONE LDS round trip for what everybody reads: loads behind a branch would each be one of their own)
    int f_abort = pg_flag(sflag), f_end = pg_flag(sflag + 1);
    double beta = bc[0];
    pg_keep(f_abort, f_end, beta);
    if (f_abort != 0) return;
    if (f_end != 0) {
      if (roww) finish();
      return;
    }
    const S bs2 = S(beta);
    stamp(2);
    product(bs2);
    __syncthreads();  // #2: the quad sums are written
    stamp(3);
    rowsums(0, bs2);
    __syncthreads();  // #3: the row sums are written
    // ---- the stagers keep the direction of their column: p = z + beta p, in place (nobody reads it before the next product)
    if (stager) {
#pragma unroll
      for (int a = 0; a < 9; ++a) pst[9 * tid + a] = pg_dir(zst[9 * tid + a], bs2, pst[9 * tid + a]);
    }
    if (roww) {
      // ---- p.q: the waves' sums of their outputs' p_c q_c ---------------------------------------------------------------
      const double s0 = ((wpq[0] + wpq[1]) + (wpq[2] + wpq[3])) + ((wpq[4] + wpq[5]) + (wpq[6] + wpq[7]));
      {
        // (the record's address from an opaque lane index: formed here, not hoisted out of the iteration loop and spilled -
        //  the reload sat right in front of this store, i.e. on the path of the p.q exchange)
        unsigned pl = unsigned(lane);
        PG_OPAQUE(pl);
        if (lane < kPgReplicas) pg_rec_store(P.part_pq + size_t(pl) * G + g, pg_pack(s0, tag));
      }
      stamp(4);
      // ---- while the partial sums travel: what the step needs of the row's state, into registers (float solver, every
      //      record of the workgroup in one pass). Behind the gather the step is then nine fused multiply-adds, the row's
      //      M^-1 (one LDS round trip) and the records - not three LDS round trips in a row.
      const bool fast = NR == 3 && !refresh && NR * W.nrows <= 64 && P.series == 0;
      unsigned fl = unsigned(lane);
      PG_OPAQUE(fl);  // (or every LDS address of this path is computed ahead of the loop and kept - i.e. spilled)
      const int ft = int(fl) < NR * W.nrows ? int(fl) : 0, frow = ft / 3, fk = ft - 3 * frow, fj0 = 9 * frow + 3 * fk;
      S rr[9], qq[9], xx[3], pp[3], bb[3];
      if (fast) {
#pragma unroll
        for (int b2 = 0; b2 < 9; ++b2) {
          rr[b2] = rs[9 * frow + b2];
          qq[b2] = qss[9 * frow + b2];
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          xx[u] = xs[fj0 + u];
          pp[u] = pcs[fj0 + u];
          bb[u] = bs[fj0 + u];
        }
      }
      // ---- exchange 2: partial sums of p.q (four workgroups per lane) ---------------------------------------------------
      pg_u32 spins = 0;
      double a_pq = 0.0;
      for (;;) {
        unsigned l0 = unsigned(lane);
        PG_OPAQUE(l0);
        const pg_rec* base = P.part_pq + size_t(g % kPgReplicas) * G;
        pg_rec r[4];
        // (clamped, masked below: lanes beyond G re-read the last record)
        const int last = G - 1;
        const int i0 = min(int(l0), last), i1 = min(int(l0) + 64, last), i2 = min(int(l0) + 128, last), i3 = min(int(l0) + 192, last);
        pg_rec_load4(base + i0, base + i1, base + i2, base + i3, r);
        bool ok = true;
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool on = int(l0) + 64 * k < G;
          const bool good = pg_unpack(r[k], tag, v[k]);
          ok &= good || !on;
          if (!on) v[k] = 0.0;
        }
        a_pq = ((v[0] + v[1]) + (v[2] + v[3]));
        if (__ballot(!ok) == 0) break;
        const int pa = poll_again(spins);
        if (pa == 2) raise_abort();
        if (pa) break;
      }
      const double pq = wave_sum(a_pq);
      stamp(5);
      // ---- step (k_pcgs_update) ----------------------------------------------------------------------------------------------
      int stop2 = 0, term2 = 0;
      double alpha = 0.0;
      if (pq != pq) {
        stop2 = 1;  // NaN: numerical failure at once
        term2 = 2;
      } else if (pq <= 0.0 || isinf(pq)) {
        stop2 = 1;  // "Matrix is indefinite, no more progress can be made." -> NO_CONVERGENCE
      } else {
        alpha = bc[1] / pq;
        if (isinf(alpha)) {
          stop2 = 1;
          term2 = 2;
        }
      }
      if (stop2) {
        my_stop = 1;
        if (lane == 0) {
          bc[5] = pq;
          bc[6] = alpha;
          endi[0] = term2;
          endi[1] = cur;
          endi[2] = term2 == 0 ? 1 : 0;
          endi[3] = 1;
          __hip_atomic_store(sflag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else if (fast) {
        if constexpr (NR == 3) {
          const S a_s = S(alpha);
          S zc[3];
#pragma unroll
          for (int b2 = 0; b2 < 9; ++b2) rr[b2] -= a_s * qq[b2];
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            xx[u] += a_s * pp[u];
            S z = S(0);
#pragma unroll
            for (int b2 = 0; b2 < 9; ++b2) z += minv[9 * (fj0 + u) + b2] * rr[b2];
            zc[u] = z;
          }
          double acc_rho = 0.0, acc_q = 0.0;
          if (lane < NR * W.nrows) {
            pg_rec_store(P.zg + size_t(NR) * size_t(W.row0 + frow) + fk, pg_pack3(float(zc[0]), float(zc[1]), float(zc[2]), tag + pg_u32(P.tag_stride)));
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              const S r_i = fk == 0 ? rr[u] : fk == 1 ? rr[3 + u] : rr[6 + u];
              acc_rho += double(r_i) * double(zc[u]);
              acc_q -= double(xx[u]) * double(bb[u] + r_i);
              xs[fj0 + u] = xx[u];
              rs[fj0 + u] = r_i;
            }
          }
          const double s0r = wave_sum(acc_rho), s1r = wave_sum(acc_q);
          if (lane < 2 * kPgReplicas) {
            unsigned pl = unsigned(lane);
            PG_OPAQUE(pl);  // (as for the p.q record above)
            const int rep = int(pl >> 1), k2 = int(pl & 1);
            pg_rec_store(P.part_rq + (size_t(rep) * G + g) * 2 + k2, pg_pack(k2 ? s1r : s0r, tag + pg_u32(P.tag_stride)));
          }
        }
      } else {
        const S a_s = S(alpha);
        for (int j = lane; j < nout; j += 64) {
          xs[j] += a_s * pcs[j];
          if (!refresh) rs[j] -= a_s * qss[j];
        }
        // residual refresh r = b - H x (conjugate_gradient.hpp:230-235): x travels like z
        if (refresh) publish_vec(P.xg, xs, tag);
      }
      stamp(6);
    }
    if (refresh) {
      if (wave_stages) stage_vector(P.xg, tag);
      __syncthreads();  // #4: x is staged
      if (pg_flag(sflag) != 0) return;
      if (pg_flag(sflag + 1) != 0) {
        if (roww) finish();
        return;
      }
      product(S(0));
      __syncthreads();  // #5
      rowsums(1, S(0));
      __syncthreads();  // #6
      if (roww) {
        unsigned j0 = unsigned(lane);
        PG_OPAQUE(j0);  // (as at the operator switch above)
        for (int j = int(j0); j < nout; j += 64) rs[j] = bs[j] - qss[j];
      }
    }
    if (!(NR == 3 && !refresh && NR * W.nrows <= 64 && P.series == 0))
      if (!close_all(tag + pg_u32(P.tag_stride))) return;
    stamp(7);
    it = cur;
    need_test = 1;
  }
}

}  // namespace rba
