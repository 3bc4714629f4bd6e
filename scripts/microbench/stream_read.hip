// What a kernel can stream from HBM on this MI355X: the ceiling the rooflines of DESIGN.md are a fraction of is the data
// sheet's ~8 TB/s; this measures the rate plain streaming kernels reach, as context for `roofline.frac` (0.64 for the
// matrix-free product means 5.1 TB/s).
//   read   sum of a 700 MB buffer (the product's size), 16-byte loads, UNR loads in flight per lane, grid-stride over
//          persistent workgroups; one value per workgroup written at the end
//   copy   16-byte loads and stores of 350 MB -> 350 MB
// for several launch shapes; best of 20 launches after 3 warm-up launches, HIP events around each launch.
//   hipcc --offload-arch=gfx950 -O3 -o stream_read.bin stream_read.hip && ./stream_read.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e = (x);                                                                   \
    if (e != hipSuccess) {                                                                \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e)); \
      std::exit(1);                                                                       \
    }                                                                                     \
  } while (0)

template <int UNR>
__global__ void k_read(const float4* __restrict__ src, size_t n4, float* __restrict__ out) {
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (; i + (UNR - 1) * stride < n4; i += UNR * stride) {
    float4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
  }
  for (; i < n4; i += stride) {
    const float4 v = src[i];
    acc += (v.x + v.y) + (v.z + v.w);
  }
  // (keeps the loads alive; one store per lane of the first wavefront would do - the cost is nothing at this size)
  if (acc == 123456.789f) out[blockIdx.x] = acc;
}

template <int UNR>
__global__ void k_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  for (; i + (UNR - 1) * stride < n4; i += UNR * stride) {
    float4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNR; ++u) dst[i + u * stride] = v[u];
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}

template <class F>
static double best_us(F&& launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  double best = 1e30;
  for (int it = 0; it < 23; ++it) {
    CHECK(hipEventRecord(e0, 0));
    launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 3 && ms * 1e3 < best) best = ms * 1e3;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::printf("%s: %d CUs\n", prop.gcnArchName, cus);
  const size_t bytes = size_t(700) << 20, n4 = bytes / 16;
  float4 *src, *dst;
  float* out;
  CHECK(hipMalloc(&src, bytes));
  CHECK(hipMalloc(&dst, bytes / 2));
  CHECK(hipMalloc(&out, 1 << 20));
  CHECK(hipMemset(src, 1, bytes));
  for (int threads : {256, 512, 1024})
    for (int per_cu : {1, 2, 4, 8}) {
      if (threads * per_cu > 2048) continue;
      const int grid = cus * per_cu;
      const double t1 = best_us([&] { hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(threads), 0, 0, src, n4, out); });
      const double t4 = best_us([&] { hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(threads), 0, 0, src, n4, out); });
      const double t8 = best_us([&] { hipLaunchKernelGGL(k_read<8>, dim3(grid), dim3(threads), 0, 0, src, n4, out); });
      std::printf("read  %4d threads x %d workgroups / CU: 1 load in flight %6.1f us = %5.2f TB/s, 4: %6.1f us = %5.2f TB/s, "
                  "8: %6.1f us = %5.2f TB/s\n",
                  threads, per_cu, t1, bytes / t1 * 1e-6, t4, bytes / t4 * 1e-6, t8, bytes / t8 * 1e-6);
    }
  // one workgroup per 64 KB as an ordinary (non-persistent) grid
  {
    const int threads = 256;
    const int grid = int((n4 + threads * 16 - 1) / (threads * 16));
    const double t = best_us([&] { hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(threads), 0, 0, src, n4, out); });
    std::printf("read  %d workgroups of 256 threads, 16 loads per lane: %6.1f us = %5.2f TB/s\n", grid, t, bytes / t * 1e-6);
  }
  const size_t h4 = n4 / 2;
  for (int threads : {256, 1024})
    for (int per_cu : {1, 2, 4}) {
      if (threads * per_cu > 2048) continue;
      const int grid = cus * per_cu;
      const double t4 = best_us([&] { hipLaunchKernelGGL(k_copy<4>, dim3(grid), dim3(threads), 0, 0, src, dst, h4); });
      std::printf("copy  %4d threads x %d workgroups / CU, 4 in flight: %6.1f us = %5.2f TB/s (read + written)\n", threads,
                  per_cu, t4, double(bytes) / t4 * 1e-6);
    }
  return 0;
}
