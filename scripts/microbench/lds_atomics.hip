// Throughput of LDS atomics with random addresses (development microbenchmark).
// One 1024-thread workgroup per CU, every lane issues N atomics to pseudo-random words of a 64 KB (u32/f32)
// or 128 KB (u64/f64) LDS array. Prints nanoseconds per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <class T> __device__ void lds_add(T* p, T v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <class T, int MODE>
__global__ __launch_bounds__(1024) void k(T* out, int n_words, int iters) {
  extern __shared__ __align__(16) unsigned char raw[];
  T* a = reinterpret_cast<T*>(raw);
  for (int i = threadIdx.x; i < n_words; i += 1024) a[i] = T(0);
  __syncthreads();
  uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    uint32_t idx;
    if (MODE == 0) idx = (s >> 8) % uint32_t(n_words);                           // fully random word
    else idx = (((s >> 8) % uint32_t(n_words / 9)) * 9 + (it % 9));              // random camera, component it%9
    lds_add(a + idx, T(1));
  }
  __syncthreads();
  T acc = T(0);
  for (int i = threadIdx.x; i < n_words; i += 1024) acc += a[i];
  if (acc == T(123457)) out[0] = acc;
}
template <class T, int MODE> void run(const char* name, int n_words) {
  T* out; hipMalloc(&out, 64);
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<T, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<T, MODE>), dim3(256), dim3(1024), n_words * sizeof(T), 0, out, n_words, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep == 1) printf("%-28s mode %d: %.3f ms, %.1f ns per wave-instruction per CU (16 waves x %d)\n", name, MODE, ms, ms * 1e6 / (16.0 * iters), iters);
  }
  hipFree(out);
}
int main() {
  run<float, 0>("ds_add_f32", 16384); run<float, 1>("ds_add_f32", 16002);
  run<unsigned, 0>("ds_add_u32", 16384); run<unsigned, 1>("ds_add_u32", 16002);
  run<unsigned long long, 0>("ds_add_u64", 16384); run<unsigned long long, 1>("ds_add_u64", 16002);
  run<double, 0>("ds_add_f64", 16384); run<double, 1>("ds_add_f64", 16002);
  return 0;
}
