// tests/hipemu/pg_record_io.hpp - the CPU execution harness's stand-in for rootba_amd/csrc/pg_record_io.hpp (build_emu.py
// puts it in that file's place): the records of the persistent PCG kernel as two 8-byte atomics - a record CAN be torn
// here, as the kernel's check expects - and no register hints. TEST INFRASTRUCTURE ONLY.
#pragma once

#include <hip/hip_runtime.h>

namespace rba {

using pg_u32 = unsigned int;
using pg_u64 = unsigned long long;
typedef pg_u32 pg_rec __attribute__((ext_vector_type(4)));  // a 16-byte record

#define PG_OPAQUE(x) ((void)0)
__device__ __forceinline__ void pg_keep(int&, int&, double&) {}
// CPU execution harness of the tests: two 8-byte atomics per record - a record CAN be torn there, as the check expects
__device__ __forceinline__ void pg_rec_store(pg_rec* p, pg_rec v) {
  pg_u64* q = reinterpret_cast<pg_u64*>(p);
  __hip_atomic_store(q, pg_u64(v.x) | (pg_u64(v.y) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, pg_u64(v.z) | (pg_u64(v.w) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ pg_rec pg_rec_load1(const pg_rec* p) {
  const pg_u64* q = reinterpret_cast<const pg_u64*>(p);
  const pg_u64 a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const pg_u64 b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  pg_rec v;
  v.x = pg_u32(a);
  v.y = pg_u32(a >> 32);
  v.z = pg_u32(b);
  v.w = pg_u32(b >> 32);
  return v;
}
template <int N>
__device__ __forceinline__ void pg_rec_load(const pg_rec* p, int stride, pg_rec (&v)[N]) {
  for (int i = 0; i < N; ++i) v[i] = pg_rec_load1(p + i * stride);
}
__device__ __forceinline__ void pg_rec_load_pairs(const pg_rec* p0, const pg_rec* p1, const pg_rec* p2, const pg_rec* p3,
                                                  pg_rec (&v)[8]) {
  const pg_rec* p[4] = {p0, p1, p2, p3};
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = pg_rec_load1(p[i]);
    v[2 * i + 1] = pg_rec_load1(p[i] + 1);
  }
}
__device__ __forceinline__ void pg_rec_load4(const pg_rec* p0, const pg_rec* p1, const pg_rec* p2, const pg_rec* p3,
                                             pg_rec (&r)[4]) {
  r[0] = pg_rec_load1(p0);
  r[1] = pg_rec_load1(p1);
  r[2] = pg_rec_load1(p2);
  r[3] = pg_rec_load1(p3);
}

}  // namespace rba
