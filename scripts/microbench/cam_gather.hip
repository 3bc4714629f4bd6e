// Memory side of the camera-major stage-2 pass (development microbenchmark, DESIGN.md 10 "open"):
// gather of a 72-byte Jacobian record + a 32-byte W8 record per observation
//   A) one workgroup per CAMERA walking that camera's observation list (what k_cam_stage2_w8_mfma does);
//   B) one workgroup per BLOCK of B consecutive cameras walking the sorted union of their lists, so that the
//      records of a landmark seen by several cameras of the block are fetched as one contiguous piece.
// Synthetic topology like the venice stand-in after the camera sort: n_lms landmarks with K = 5 observations each,
// stored landmark-major, landmark l seen by cameras f, f+1, ..., f+K-1 with f ascending in l.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int RW = 26, CH = 32;
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ J, const float* __restrict__ W,
                                                const int* __restrict__ list, const int64_t* __restrict__ off,
                                                float* __restrict__ out, int n_groups) {
  __shared__ __attribute__((aligned(16))) float stage[4][CH * RW + 6];
  const int per = (n_groups + 7) / 8;
  const int c = (blockIdx.x % 8) * per + blockIdx.x / 8;  // XCD-contiguous groups
  if (c >= n_groups) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t t0 = off[c], t1 = off[c + 1];
  float* lds = stage[wave];
  float acc = 0.f;
  for (int64_t base = t0 + CH * wave; base < t1; base += 4 * CH) {
    const int cnt = int(t1 - base < CH ? t1 - base : CH);
    const int idx = lane < cnt ? list[base + lane] : 0;
#pragma unroll
    for (int j = 0; j < (CH * 9 + 63) / 64; ++j) {
      const int q = j * 64 + lane, r = q / 9, pc = q - 9 * r;
      const int o = __shfl(idx, r & 31);
      if (q < cnt * 9) *reinterpret_cast<float2*>(lds + r * RW + 2 * pc) = *reinterpret_cast<const float2*>(J + int64_t(o) * 18 + 2 * pc);
    }
    {
      const int r = lane >> 1, h = lane & 1;
      const int o = __shfl(idx, r & 31);
      if (r < cnt) {
        const float4 w = *reinterpret_cast<const float4*>(W + int64_t(o) * 8 + 4 * h);
        float* d = lds + r * RW + 18 + 4 * h;
        *reinterpret_cast<float2*>(d) = float2{w.x, w.y};
        *reinterpret_cast<float2*>(d + 2) = float2{w.z, w.w};
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int s = lane; s < cnt * RW; s += 64) acc += lds[s];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (acc == 1.2345e30f) out[0] = acc;
}

// Variant: no merged lists. A wavefront owns quarter q of ONE camera's list (same inner loop as the product kernel);
// a workgroup is the same quarter of FOUR ADJACENT cameras, whose lists cover the same landmarks in the same order,
// so the four waves touch the same cache lines at about the same time.
__global__ __launch_bounds__(256) void k_gather_adjacent(const float* __restrict__ J, const float* __restrict__ W,
                                                         const int* __restrict__ list, const int64_t* __restrict__ qoff,
                                                         float* __restrict__ out, int n_cams) {
  __shared__ __attribute__((aligned(16))) float stage[4][CH * RW + 6];
  const int n_wg = ((n_cams + 3) / 4) * 4;  // (camera block, quarter)
  const int per = (n_wg + 7) / 8;
  const int wg = (blockIdx.x % 8) * per + blockIdx.x / 8;
  if (wg >= n_wg) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cam = (wg / 4) * 4 + wave, q = wg % 4;
  if (cam >= n_cams) return;
  const int64_t t0 = qoff[4 * cam + q], t1 = qoff[4 * cam + q + 1];
  float* lds = stage[wave];
  float acc = 0.f;
  for (int64_t base = t0; base < t1; base += CH) {
    const int cnt = int(t1 - base < CH ? t1 - base : CH);
    const int idx = lane < cnt ? list[base + lane] : 0;
#pragma unroll
    for (int j = 0; j < (CH * 9 + 63) / 64; ++j) {
      const int qq = j * 64 + lane, r = qq / 9, pc = qq - 9 * r;
      const int o = __shfl(idx, r & 31);
      if (qq < cnt * 9) *reinterpret_cast<float2*>(lds + r * RW + 2 * pc) = *reinterpret_cast<const float2*>(J + int64_t(o) * 18 + 2 * pc);
    }
    {
      const int r = lane >> 1, h = lane & 1;
      const int o = __shfl(idx, r & 31);
      if (r < cnt) {
        const float4 w = *reinterpret_cast<const float4*>(W + int64_t(o) * 8 + 4 * h);
        float* d = lds + r * RW + 18 + 4 * h;
        *reinterpret_cast<float2*>(d) = float2{w.x, w.y};
        *reinterpret_cast<float2*>(d + 2) = float2{w.z, w.w};
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int s = lane; s < cnt * RW; s += 64) acc += lds[s];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (acc == 1.2345e30f) out[0] = acc;
}

int main() {
  const int n_cams = 1778, K = 5;
  const int n_lms = 1000000;
  const int64_t n_obs = int64_t(n_lms) * K;
  std::vector<std::vector<int>> cam_list(n_cams);
  for (int l = 0; l < n_lms; ++l) {
    const int f = int(int64_t(l) * n_cams / n_lms);
    for (int i = 0; i < K; ++i) cam_list[(f + i) % n_cams].push_back(l * K + i);
  }
  float *J, *W, *out;
  CK(hipMalloc(&J, n_obs * 18 * 4)); CK(hipMalloc(&W, n_obs * 8 * 4)); CK(hipMalloc(&out, 64));
  CK(hipMemset(J, 0, n_obs * 18 * 4)); CK(hipMemset(W, 0, n_obs * 8 * 4));
  // (a workgroup per whole block starves the chip: 445 / 223 / 112 / 56 workgroups run 370 / 640 / 1271 / 2539 us;
  //  so the merged list of a block is cut into B equal SEGMENTS, one workgroup each - the partial sums per camera
  //  would have to be combined afterwards)
  for (int B : {1, 4, 8, 16, 32, 64}) {
    const int n_blocks = (n_cams + B - 1) / B;
    std::vector<int> list; std::vector<int64_t> off(1, 0);
    for (int g = 0; g < n_blocks; ++g) {
      std::vector<int> u;
      const int nb = std::min(n_cams, (g + 1) * B) - g * B;
      for (int c = g * B; c < g * B + nb; ++c) u.insert(u.end(), cam_list[c].begin(), cam_list[c].end());
      std::sort(u.begin(), u.end());
      for (int sgm = 0; sgm < nb; ++sgm) {
        const size_t a = u.size() * sgm / nb, b = u.size() * (sgm + 1) / nb;
        list.insert(list.end(), u.begin() + a, u.begin() + b);
        off.push_back(int64_t(list.size()));
      }
    }
    const int n_groups = int(off.size()) - 1;
    int* dl; int64_t* doff;
    CK(hipMalloc(&dl, list.size() * 4)); CK(hipMalloc(&doff, off.size() * 8));
    CK(hipMemcpy(dl, list.data(), list.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(doff, off.data(), off.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_gather, dim3(8 * ((n_groups + 7) / 8)), dim3(256), 0, 0, J, W, dl, doff, out, n_groups);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0) best = std::min(best, ms);
    }
    printf("blocks of %2d cameras, one segment per workgroup (%4d workgroups): %.1f us, %.2f TB/s of the %.0f MB needed\n", B, n_groups,
           best * 1e3, n_obs * 104.0 / best * 1e-9, n_obs * 104.0 * 1e-6);
    CK(hipFree(dl)); CK(hipFree(doff));
  }
  {
    // per-camera lists, cut into four quarters (a wave each)
    std::vector<int> list; std::vector<int64_t> qoff(size_t(4) * n_cams + 1, 0);
    for (int c = 0; c < n_cams; ++c) {
      const auto& u = cam_list[c];
      for (int q = 0; q < 4; ++q) {
        const size_t a = u.size() * q / 4, b = u.size() * (q + 1) / 4;
        list.insert(list.end(), u.begin() + a, u.begin() + b);
        qoff[size_t(4) * c + q + 1] = int64_t(list.size());
      }
    }
    int* dl; int64_t* doff;
    CK(hipMalloc(&dl, list.size() * 4)); CK(hipMalloc(&doff, qoff.size() * 8));
    CK(hipMemcpy(dl, list.data(), list.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(doff, qoff.data(), qoff.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    const int n_wg = ((n_cams + 3) / 4) * 4;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_gather_adjacent, dim3(8 * ((n_wg + 7) / 8)), dim3(256), 0, 0, J, W, dl, doff, out, n_cams);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0) best = std::min(best, ms);
    }
    printf("wave = quarter of one camera, workgroup = the same quarter of 4 adjacent cameras (%d workgroups): %.1f us, %.2f TB/s\n",
           n_wg, best * 1e3, n_obs * 104.0 / best * 1e-9);
  }
  return 0;
}
