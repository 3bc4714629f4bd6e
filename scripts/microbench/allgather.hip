// All-gather of one double per workgroup inside ONE launch on gfx950 - the exchange behind each of the two dot products
// of a PCG iteration in kernels_pcgp.hpp - in several forms, timed with the chip-wide 100 MHz counter (s_memrealtime):
// per round every workgroup waits a pseudo-random 0 .. `skew` ns ("its own work"), publishes, and one wavefront polls
// until it has everybody's value of THIS round. Reported: ns per round and, per round, from the LAST publication to the
// first / median / last workgroup holding the complete sum.
//   0  8-byte {word, tag} granules, two per double, [G][2]; one wavefront sweeps all (2 G / 64 loads per lane)
//   1  16-byte {double, tag, check} records, one 16-byte store / load each (check = tag ^ hi ^ lo: a torn record fails it)
//   2  form 1, every record in a 128-byte line of its own (one writer per line)
//   3  form 1, `R` replicas: the publisher stores R copies with ONE instruction, a workgroup polls replica (b % R)
//   4  form 1 polled by FOUR wavefronts (a record per lane and pass), combined through LDS
//   5  form 3 + form 4
// Every round's sum is checked.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>

using u32 = unsigned;
using u64 = unsigned long long;
#define AGENT __HIP_MEMORY_SCOPE_AGENT
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u64 ld64(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, AGENT); }
__device__ __forceinline__ void st64(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, AGENT); }
// 16-byte write-through store / L1-bypassing load (sc1)
__device__ __forceinline__ void st128(u32x4* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
// (four loads in flight, ONE wait: an asm load is invisible to the compiler's wait-count bookkeeping)
__device__ __forceinline__ void ld128x4(const u32x4* p0, const u32x4* p1, const u32x4* p2, const u32x4* p3, u32x4& v0,
                                        u32x4& v1, u32x4& v2, u32x4& v3) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
      "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
      : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
      : "memory");
}
__device__ __forceinline__ u32x4 ld128(const u32x4* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
constexpr u32 kSpinLimit = 1u << 22;
constexpr int kMaxG = 256;

template <int FORM>
__global__ __launch_bounds__(512) void k(u64* gran, u32x4* rec, int R, int iters, int G, int skew_ticks, long long* t_pub,
                                         long long* t_done, u32* bad) {
  __shared__ double part[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  u32 nbad = 0;
  u32 seed = b * 2654435761u + 12345u;
  const int stride = FORM == 2 ? 8 : 1;  // records per publisher slot (8 x 16 B = a line)
  for (int it = 1; it <= iters; ++it) {
    const u32 tag = u32(it);
    const double mine = double(b + 1) * double(it);
    const double expect = 0.5 * double(G) * double(G + 1) * double(it);
    // "own work" of pseudo-random length
    seed = seed * 1664525u + 1013904223u;
    if (skew_ticks > 0 && tid == 0) {
      const long long until = wall_clock64() + (long long)((seed >> 8) % u32(skew_ticks));
      while (wall_clock64() < until) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    // ---- publish ----
    if (tid == 0 && it <= 64) t_pub[size_t(b) * 64 + (it - 1)] = wall_clock64();
    const u64 bits = u64(__double_as_longlong(mine));
    if (FORM == 0) {
      u64* gr = gran + size_t(it & 1) * 2 * kMaxG;
      if (tid == 0) {
        st64(gr + 2 * b, (u64(tag) << 32) | (bits >> 32));
        st64(gr + 2 * b + 1, (u64(tag) << 32) | (bits & 0xffffffffull));
      }
    } else {
      const u32 hi = u32(bits >> 32), lo = u32(bits);
      const u32x4 v = {lo, hi, tag, tag ^ hi ^ lo};
      const int reps = (FORM == 3 || FORM == 5) ? R : 1;
      u32x4* base = rec + size_t(it & 1) * size_t(kMaxG) * 8 * 32;
      if (tid < reps) st128(base + (size_t(tid) * kMaxG + b) * stride, v);
    }
    // ---- poll ----
    double total = 0.0;
    bool fine = true;
    const int pollers = (FORM == 4 || FORM == 5) ? 4 : 1;
    if (wave < pollers) {
      u32 spins = 0;
      double acc = 0.0;
      for (;;) {
        bool ok = true;
        acc = 0.0;
        if (FORM == 0) {
          const u64* gr = gran + size_t(it & 1) * 2 * kMaxG;
          for (int g = lane; g < G; g += 64) {
            const u64 h = ld64(gr + 2 * g), l = ld64(gr + 2 * g + 1);
            ok &= u32(h >> 32) == tag && u32(l >> 32) == tag;
            acc += __longlong_as_double((long long)((h << 32) | (l & 0xffffffffull)));
          }
        } else {
          const int rep = (FORM == 3 || FORM == 5) ? b % R : 0;
          const u32x4* base = rec + size_t(it & 1) * size_t(kMaxG) * 8 * 32 + size_t(rep) * kMaxG * stride;
          auto take = [&](const u32x4& v, bool on) {
            if (on) {
              ok &= v.z == tag && v.w == (tag ^ v.y ^ v.x);
              acc += __longlong_as_double((long long)((u64(v.y) << 32) | u64(v.x)));
            }
          };
          if (pollers == 4) {
            const int g = lane + 64 * wave;
            const u32x4 v = ld128(base + size_t(min(g, G - 1)) * stride);
            take(v, g < G);
          } else {
            u32x4 v0, v1, v2, v3;
            ld128x4(base + size_t(min(lane, G - 1)) * stride, base + size_t(min(lane + 64, G - 1)) * stride,
                    base + size_t(min(lane + 128, G - 1)) * stride, base + size_t(min(lane + 192, G - 1)) * stride, v0, v1, v2,
                    v3);
            take(v0, lane < G);
            take(v1, lane + 64 < G);
            take(v2, lane + 128 < G);
            take(v3, lane + 192 < G);
          }
        }
        if (__ballot(!ok) == 0) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinLimit) {
          fine = false;
          break;
        }
      }
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      if (lane == 0) part[wave] = acc;
    }
    __syncthreads();
    total = part[0];
    for (int w = 1; w < pollers; ++w) total += part[w];
    if (tid == 0 && it <= 64) t_done[size_t(b) * 64 + (it - 1)] = wall_clock64();
    if (!fine || total != expect) ++nbad;
    __syncthreads();
  }
  if (tid == 0 && nbad) atomicAdd(bad, nbad);
}

template <int FORM>
void run(const char* name, int G, int R, int skew_ns, int iters, u64* gran, u32x4* rec, long long* t_pub, long long* t_done,
         u32* bad) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  u32 hbad = 0;
  std::vector<long long> hp(size_t(kMaxG) * 64), hd(size_t(kMaxG) * 64);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(gran, 0, 4 * kMaxG * sizeof(u64));
    hipMemset(rec, 0, size_t(2) * kMaxG * 8 * 32 * sizeof(u32x4));
    hipMemset(bad, 0, sizeof(u32));
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<FORM>), dim3(G), dim3(512), 0, 0, gran, rec, R, iters, G, skew_ns / 10, t_pub, t_done, bad);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
    u32 t;
    hipMemcpy(&t, bad, sizeof t, hipMemcpyDeviceToHost);
    hbad += t;
  }
  hipMemcpy(hp.data(), t_pub, hp.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hd.data(), t_done, hd.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> first, med, last;
  for (int i = 8; i < 64; ++i) {
    long long lp = 0;
    std::vector<long long> d;
    for (int b = 0; b < G; ++b) lp = std::max(lp, hp[size_t(b) * 64 + i]);
    for (int b = 0; b < G; ++b) d.push_back(hd[size_t(b) * 64 + i] - lp);
    std::sort(d.begin(), d.end());
    first.push_back(d.front() * 10.0);
    med.push_back(d[d.size() / 2] * 10.0);
    last.push_back(d.back() * 10.0);
  }
  auto median = [](std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
  };
  printf("form %d G %3d R %2d skew %4d ns  %-46s %7.0f ns / round; last publication -> complete: first %5.0f median %5.0f last %5.0f ns (bad %u)\n",
         FORM, G, R, skew_ns, name, best * 1e6 / iters, median(first), median(med), median(last), hbad);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  u64* gran;
  u32x4* rec;
  long long *t_pub, *t_done;
  u32* bad;
  hipMalloc(&gran, 4 * kMaxG * sizeof(u64));
  hipMalloc(&rec, size_t(2) * kMaxG * 8 * 32 * sizeof(u32x4));
  hipMalloc(&t_pub, size_t(kMaxG) * 64 * 8);
  hipMalloc(&t_done, size_t(kMaxG) * 64 * 8);
  hipMalloc(&bad, sizeof(u32));
  for (int G : {230, 64}) {
    for (int skew : {0, 1500}) {
      run<0>("8-byte granules, one wave sweeps", G, 1, skew, iters, gran, rec, t_pub, t_done, bad);
      run<1>("16-byte records", G, 1, skew, iters, gran, rec, t_pub, t_done, bad);
      run<2>("16-byte records, a line each", G, 1, skew, iters, gran, rec, t_pub, t_done, bad);
      run<3>("16-byte records, 8 replicas", G, 8, skew, iters, gran, rec, t_pub, t_done, bad);
      run<3>("16-byte records, 32 replicas", G, 32, skew, iters, gran, rec, t_pub, t_done, bad);
      run<4>("16-byte records, four polling waves", G, 1, skew, iters, gran, rec, t_pub, t_done, bad);
      run<5>("16-byte records, 8 replicas, four polling waves", G, 8, skew, iters, gran, rec, t_pub, t_done, bad);
    }
  }
  return 0;
}
