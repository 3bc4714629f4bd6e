// pg_record_io.hpp - the memory primitives of the persistent PCG kernel (kernels_pcgp.hpp): 16-byte self-validating
// records written by ONE write-through store and read past L1 (`sc1`), as inline gfx950 assembly - no builtin emits a
// 16-byte access of agent scope - and the register-allocation hints of that kernel. (The CPU execution harness of the
// tests substitutes its own file of the same name, tests/hipemu/pg_record_io.hpp: this one has a single code path.)
#pragma once

#include <hip/hip_runtime.h>

namespace rba {

using pg_u32 = unsigned int;
using pg_u64 = unsigned long long;
typedef pg_u32 pg_rec __attribute__((ext_vector_type(4)));  // a 16-byte record

// The kernel keeps 162 registers of matrix per lane; every loop-invariant address the compiler hoists out of the
// iteration loop (a record address per exchange and lane ...) is a spilled pair. An index made opaque at its use is
// recomputed there (one multiply-add) instead.
#define PG_OPAQUE(x) asm volatile("" : "+v"(x))
// values read from LDS right behind a workgroup barrier, pinned to registers before the branches that use them
__device__ __forceinline__ void pg_keep(int& a, int& b, double& c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); }

// ---- 16-byte records: write-through store, L1-bypassing loads (sc1) ---------------------------------------------------
// (inline assembly: no builtin emits a 16-byte access of agent scope. The loads of a sweep and their wait are ONE
//  statement - the compiler does not count the memory operations of an asm statement.)
__device__ __forceinline__ void pg_rec_store(pg_rec* p, pg_rec v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
// N records at p, p + stride, ...: all loads in flight, one wait
__device__ __forceinline__ void pg_rec_load(const pg_rec* p, int, pg_rec (&v)[1]) {
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v[0]) : "v"(p) : "memory");
}
__device__ __forceinline__ void pg_rec_load(const pg_rec* p, int stride, pg_rec (&v)[2]) {
  const pg_rec* p1 = p + stride;
  asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1])
               : "v"(p), "v"(p1)
               : "memory");
}
__device__ __forceinline__ void pg_rec_load(const pg_rec* p, int stride, pg_rec (&v)[3]) {
  const pg_rec *p1 = p + stride, *p2 = p + 2 * stride;
  asm volatile(
      "global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2])
      : "v"(p), "v"(p1), "v"(p2)
      : "memory");
}
__device__ __forceinline__ void pg_rec_load(const pg_rec* p, int stride, pg_rec (&v)[4]) {
  const pg_rec *p1 = p + stride, *p2 = p + 2 * stride, *p3 = p + 3 * stride;
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
      "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
      : "v"(p), "v"(p1), "v"(p2), "v"(p3)
      : "memory");
}
// four PAIRS of adjacent records (the rho and Q partial sums of four workgroups)
__device__ __forceinline__ void pg_rec_load_pairs(const pg_rec* p0, const pg_rec* p1, const pg_rec* p2, const pg_rec* p3,
                                                  pg_rec (&v)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %9, off sc1\n\tglobal_load_dwordx4 %3, %9, off offset:16 sc1\n\t"
      "global_load_dwordx4 %4, %10, off sc1\n\tglobal_load_dwordx4 %5, %10, off offset:16 sc1\n\t"
      "global_load_dwordx4 %6, %11, off sc1\n\tglobal_load_dwordx4 %7, %11, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
      : "memory");
}
__device__ __forceinline__ void pg_rec_load(const pg_rec* p, int stride, pg_rec (&v)[9]) {
  // (the double solver's vectors: nine records per camera, immediate offsets of 16 bytes - stride is 1)
  asm volatile(
      "global_load_dwordx4 %0, %9, off sc1\n\tglobal_load_dwordx4 %1, %9, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %9, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %9, off offset:48 sc1\n\t"
      "global_load_dwordx4 %4, %9, off offset:64 sc1\n\tglobal_load_dwordx4 %5, %9, off offset:80 sc1\n\t"
      "global_load_dwordx4 %6, %9, off offset:96 sc1\n\tglobal_load_dwordx4 %7, %9, off offset:112 sc1\n\t"
      "global_load_dwordx4 %8, %9, off offset:128 sc1\n\ts_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8])
      : "v"(p)
      : "memory");
}
// four records at four addresses: all loads in flight, one wait
__device__ __forceinline__ void pg_rec_load4(const pg_rec* p0, const pg_rec* p1, const pg_rec* p2, const pg_rec* p3,
                                             pg_rec (&r)[4]) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off "
      "sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
      : "memory");
}

}  // namespace rba
