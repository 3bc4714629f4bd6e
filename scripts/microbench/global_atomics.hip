// Throughput of device-scope global atomics with random addresses (development microbenchmark):
// 2048 workgroups x 256 threads, every lane issues N atomics "9 consecutive lanes -> 9 consecutive words
// of a random camera" into a vector of n_words (the scatter pattern of the matrix-free product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <class T> __device__ void g_add(T* p, T v) { unsafeAtomicAdd(p, v); }
__device__ void g_add(unsigned* p, unsigned v) { atomicAdd(p, v); }
__device__ void g_add(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
template <class T>
__global__ __launch_bounds__(256) void k(T* y, int n_cams, int iters) {
  const int lane = threadIdx.x & 63;
  uint32_t s = (blockIdx.x * 256 + (threadIdx.x - lane) + lane / 9) * 2654435761u + 12345u;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const uint32_t cam = (s >> 8) % uint32_t(n_cams);
    if (lane < 63) g_add(y + 9 * cam + lane % 9, T(1));
  }
}
template <class T> void run(const char* name, int n_cams) {
  T* y; hipMalloc(&y, size_t(9) * n_cams * sizeof(T)); hipMemset(y, 0, size_t(9) * n_cams * sizeof(T));
  const int iters = 100, wgs = 2048;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<T>), dim3(wgs), dim3(256), 0, 0, y, n_cams, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = double(wgs) * 252 * iters;
    if (rep == 1) printf("%-24s n_cams %6d: %.3f ms, %.1f G lane-atomics/s (venice product = 45 M -> %.0f us)\n", name, n_cams, ms, ops / ms * 1e-6, 45e6 / (ops / ms) * 1e3);
  }
  hipFree(y);
}
int main() {
  for (int n : {1778, 13682}) {
    run<float>("global_atomic_add_f32", n); run<double>("global_atomic_add_f64", n);
    run<unsigned>("global_atomic_add_u32", n); run<unsigned long long>("global_atomic_add_u64", n);
  }
  return 0;
}
