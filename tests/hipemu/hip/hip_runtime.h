// tests/hipemu/hip/hip_runtime.h - a CPU EXECUTION HARNESS for the kernels of rootba_amd/csrc (development / test tool).
//
// TEST INFRASTRUCTURE ONLY. tests/hipemu/build_emu.py compiles the UNCHANGED product sources (solver.hip + kernels*.hpp)
// as plain C++ against this header into tests/hipemu/_build/librootba_hip_emu.so: every kernel then runs on the CPU with
// the gfx950 execution model emulated - a workgroup is a set of cooperative fibers, one per work-item, 64 per wavefront;
// __syncthreads, the wave-level operations the kernels use (DPP moves, readlane, __shfl, the 16x16x4 f32 matrix-core
// instruction) and LDS are modelled; workgroups of a grid run one after the other. It exists so that kernel LOGIC can be
// exercised when no GPU is at hand (GPU time is rationed to minutes per round). It is NOT a backend of the product:
// nothing under rootba_amd/ or include/ refers to it, the product library still needs a GPU and fails loudly without one,
// and no parity or performance statement rests on it - rounding differs from the hardware (fma contraction, atomics
// order) and it is orders of magnitude slower. tests select it with RBA_EMU=1 (tests/conftest.py).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>

// ---- the library dlopens "librccl.so.1"; a process that has imported torch already holds torch's own RCCL under that
// name, so on the harness the stand-in (fake_rccl.cpp) is opened by the path in HIPEMU_RCCL instead
#include <dlfcn.h>
static inline void* hipemu_dlopen(const char* name, int flags) {
  const char* fake = std::getenv("HIPEMU_RCCL");
  if (fake && name && std::strncmp(name, "librccl", 7) == 0) return dlopen(fake, RTLD_NOW | RTLD_LOCAL);
  return dlopen(name, flags);
}
#define dlopen hipemu_dlopen

// ---- qualifiers --------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __align__(n) __attribute__((aligned(n)))
// static __shared__ arrays are function-level thread-local statics shared by the fibers of the workgroup that runs on this
// OS thread (one workgroup at a time per thread); `extern __shared__` declarations are rewritten to `extern thread_local`
// by build_emu.py and defined in the runtime
#define __shared__ static thread_local

// ---- vector types --------------------------------------------------------------------------------------------------
// HIPEMU_STRICT_ALIGN (the UBSan build, HIPEMU_SANITIZE=alignment,bounds): the alignments HIP gives these types, so that
// `-fsanitize=alignment` reports every vector access whose address is not a multiple of the vector size (the LDS
// instructions for 8 / 16 bytes want that; vector accesses to global memory only need 4)
#ifdef HIPEMU_STRICT_ALIGN
// (user-provided copy operations: a trivial aggregate copy is emitted as a memcpy, which the sanitizer does not check;
//  a call of a member function on a misaligned object is)
#define HIPEMU_VEC2(NAME, T, A)                                              \
  struct alignas(A) NAME {                                                   \
    T x, y;                                                                  \
    NAME() = default;                                                        \
    NAME(T a, T b) : x(a), y(b) {}                                           \
    NAME(const NAME& o) : x(o.x), y(o.y) {}                                  \
    NAME& operator=(const NAME& o) { x = o.x; y = o.y; return *this; }       \
  };
#define HIPEMU_VEC4(NAME, T, A)                                                              \
  struct alignas(A) NAME {                                                                   \
    T x, y, z, w;                                                                            \
    NAME() = default;                                                                        \
    NAME(T a, T b, T c, T d) : x(a), y(b), z(c), w(d) {}                                     \
    NAME(const NAME& o) : x(o.x), y(o.y), z(o.z), w(o.w) {}                                  \
    NAME& operator=(const NAME& o) { x = o.x; y = o.y; z = o.z; w = o.w; return *this; }     \
  };
#else
#define HIPEMU_VEC2(NAME, T, A) struct NAME { T x, y; };
#define HIPEMU_VEC4(NAME, T, A) struct NAME { T x, y, z, w; };
#endif
HIPEMU_VEC2(float2, float, 8)
HIPEMU_VEC4(float4, float, 16)
HIPEMU_VEC2(double2, double, 16)
HIPEMU_VEC4(double4, double, 32)
HIPEMU_VEC2(int2, int, 8)
HIPEMU_VEC4(int4, int, 16)
HIPEMU_VEC2(uint2, unsigned, 8)
HIPEMU_VEC4(uint4, unsigned, 16)
struct uint3 { unsigned x, y, z; };
struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
static inline double4 make_double4(double x, double y, double z, double w) { return double4{x, y, z, w}; }

// ---- runtime types / constants ---------------------------------------------------------------------------------------
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorNotReady = 600;
constexpr hipError_t hipErrorInvalidValue = 1;
constexpr hipError_t hipErrorStreamCaptureUnsupported = 900;
struct hipemu_stream;
struct hipemu_event;
struct hipemu_graph;
typedef hipemu_stream* hipStream_t;
typedef hipemu_event* hipEvent_t;
typedef hipemu_graph* hipGraph_t;
typedef hipemu_graph* hipGraphExec_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1, hipDeviceAttributeWallClockRate = 2 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0;

namespace hipemu {
struct Fiber;
struct Self {
  uint3 tid, bid;
  dim3 bdim, gdim;
  int lane, wave;
};
Self* self();                  // the running work-item
void block_barrier();          // __syncthreads
void wave_barrier();           // all live lanes of the wavefront
void spin_yield();             // s_sleep in a spin loop: let the other work-items (and workgroups) run
uint64_t* wave_slots();        // 64 exchange slots of the running wavefront (+ 6 x 64 more for the matrix-core emulation)
int first_live_lane();         // lowest lane of the wavefront that has not returned
void launch(dim3 grid, dim3 block, size_t shmem, hipStream_t stream, std::function<void()> body);
}  // namespace hipemu

#define threadIdx (hipemu::self()->tid)
#define blockIdx (hipemu::self()->bid)
#define blockDim (hipemu::self()->bdim)
#define gridDim (hipemu::self()->gdim)
constexpr int warpSize = 64;

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch(dim3(grid), dim3(block), size_t(shmem), stream, [=]() { kernel(__VA_ARGS__); })

// ---- runtime API (tests/hipemu/hipemu_runtime.cpp) -----------------------------------------------------------------------
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int);
hipError_t hipGetDevice(int*);
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int dev);
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
template <class T>
hipError_t hipMalloc(T** p, size_t n) {
  return hipMalloc(reinterpret_cast<void**>(p), n);
}

hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
template <class T>
hipError_t hipHostMalloc(T** p, size_t n, unsigned flags = 0) {
  return hipHostMalloc(reinterpret_cast<void**>(p), n, flags);  // (the non-template overload, declared above)
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind k);
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t s);
hipError_t hipMemset(void* dst, int v, size_t n);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void* f, int block, size_t lds);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode m);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g);
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t);
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s);
hipError_t hipGraphExecDestroy(hipGraphExec_t e);
hipError_t hipGraphDestroy(hipGraph_t g);
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
template <class F>
hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) {
  return hipSuccess;
}

// ---- device-side functions ---------------------------------------------------------------------------------------------
static inline void __syncthreads() { hipemu::block_barrier(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline int __float_as_int(float v) { int r; std::memcpy(&r, &v, 4); return r; }
static inline float __int_as_float(int v) { float r; std::memcpy(&r, &v, 4); return r; }
static inline unsigned __float_as_uint(float v) { unsigned r; std::memcpy(&r, &v, 4); return r; }
static inline float __uint_as_float(unsigned v) { float r; std::memcpy(&r, &v, 4); return r; }
static inline long long __double_as_longlong(double v) { long long r; std::memcpy(&r, &v, 8); return r; }
static inline double __longlong_as_double(long long v) { double r; std::memcpy(&r, &v, 8); return r; }
// correctly rounded single operations that must not be contracted into an fma
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline double __dmul_rn(double a, double b) { volatile double r = a * b; return r; }
static inline double __dadd_rn(double a, double b) { volatile double r = a + b; return r; }
using std::abs;
using std::fabs;
using std::fma;
using std::fmaf;
using std::isfinite;
using std::isinf;
using std::isnan;
using std::max;
using std::min;
using std::sqrt;

static inline int min(int a, unsigned b) { return a < int(b) ? a : int(b); }
static inline long long min(long long a, int b) { return a < b ? a : b; }
static inline long long min(int a, long long b) { return a < b ? a : b; }
static inline long min(long a, int b) { return a < b ? a : b; }
static inline long min(int a, long b) { return a < b ? a : b; }
static inline long long max(long long a, int b) { return a > b ? a : b; }
static inline long max(long a, int b) { return a > b ? a : b; }
static inline long max(int a, long b) { return a > b ? a : b; }
static inline float rsqrtf(float v) { return 1.f / std::sqrt(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz(unsigned(v)) : 32; }

// atomics: workgroups run concurrently on several OS threads, so these are real atomic read-modify-writes
template <class T>
static inline T atomicAdd(T* p, T v) {
  if constexpr (std::is_integral<T>::value) {
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
  } else {
    T old, want;
    __atomic_load(p, &old, __ATOMIC_RELAXED);
    do {
      want = old + v;
    } while (!__atomic_compare_exchange(p, &old, &want, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return old;
  }
}
template <class T>
static inline T unsafeAtomicAdd(T* p, T v) { return atomicAdd(p, v); }
static inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline int atomicMax(int* p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
  return old;
}
static inline int atomicMin(int* p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old > v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
  return old;
}
static inline int atomicExch(int* p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
static inline int atomicCAS(int* p, int c, int v) {
  __atomic_compare_exchange_n(p, &c, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return c;
}
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
template <class T, class V>
static inline T hipemu_atomic_fetch_add(T* p, V v) { return atomicAdd(p, T(v)); }
#define __hip_atomic_fetch_add(p, v, order, scope) hipemu_atomic_fetch_add(p, v)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(p, v, __ATOMIC_RELAXED)
#define __hip_atomic_load(p, order, scope) __atomic_load_n(p, __ATOMIC_RELAXED)

// ---- wave-level operations ----------------------------------------------------------------------------------------------
namespace hipemu {
template <class T>
static inline uint64_t bits(T v) {
  static_assert(sizeof(T) <= 8, "exchange slot");
  uint64_t r = 0;
  std::memcpy(&r, &v, sizeof(T));
  return r;
}
template <class T>
static inline T from_bits(uint64_t b) {
  T r;
  std::memcpy(&r, &b, sizeof(T));
  return r;
}
// every live lane deposits `v`, then reads the slot `pick(lane)` chooses (-1: keep `fallback`)
template <class T, class F>
static inline T exchange(T v, T fallback, F pick) {
  uint64_t* s = wave_slots();
  const int lane = self()->lane;
  s[lane] = bits(v);
  wave_barrier();
  const int src = pick(lane);
  const T r = src < 0 ? fallback : from_bits<T>(s[src & 63]);
  wave_barrier();  // the slots are free for the next operation
  return r;
}
// source lane of a DPP control word (quad_perm, row_shl/shr/ror, row_mirror, row_half_mirror, row_bcast15/31);
// -1: no source (the instruction keeps `old`)
static inline int dpp_source(int ctrl, int lane) {
  const int row = lane & ~15, i = lane & 15;
  if (ctrl <= 0xff) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  if (ctrl >= 0x101 && ctrl <= 0x10f) return i + (ctrl & 15) <= 15 ? lane + (ctrl & 15) : -1;  // row_shl
  if (ctrl >= 0x111 && ctrl <= 0x11f) return i - (ctrl & 15) >= 0 ? lane - (ctrl & 15) : -1;   // row_shr
  if (ctrl >= 0x121 && ctrl <= 0x12f) return row | ((i - (ctrl & 15)) & 15);                   // row_ror
  if (ctrl == 0x140) return row | (15 - i);                                                    // row_mirror
  if (ctrl == 0x141) return row | (i & 8) | (7 - (i & 7));                                     // row_half_mirror
  if (ctrl == 0x142) return lane >= 16 ? row - 1 : -1;                                         // row_bcast:15
  if (ctrl == 0x143) return lane >= 32 ? 31 : -1;                                              // row_bcast:31
  std::fprintf(stderr, "hipemu: DPP control 0x%x is not modelled\n", ctrl);
  std::abort();
}
}  // namespace hipemu

static inline int hipemu_update_dpp(int old, int src, int ctrl) {
  return hipemu::exchange<int>(src, old, [ctrl](int lane) { return hipemu::dpp_source(ctrl, lane); });
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl) hipemu_update_dpp(old, src, ctrl)
#define __builtin_amdgcn_mov_dpp(src, ctrl, row_mask, bank_mask, bound_ctrl) hipemu_update_dpp(0, src, ctrl)
static inline int hipemu_readlane(int v, int l) {
  return hipemu::exchange<int>(v, v, [l](int) { return l; });
}
static inline int hipemu_readfirstlane(int v) {
  const int l = hipemu::first_live_lane();
  return hipemu::exchange<int>(v, v, [l](int) { return l; });
}
#define __builtin_amdgcn_readlane(v, l) hipemu_readlane(v, l)
#define __builtin_amdgcn_readfirstlane(v) hipemu_readfirstlane(v)
#define __builtin_amdgcn_wave_barrier() hipemu::wave_barrier()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_sleep(n) hipemu::spin_yield()
static inline long long clock64() { return (long long)__builtin_ia32_rdtsc(); }
static inline long long wall_clock64() { return (long long)__builtin_ia32_rdtsc(); }
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
#define amdgpu_waves_per_eu(...)  // (__attribute__((amdgpu_waves_per_eu(n))) -> __attribute__(()))
template <class T>
static inline T __shfl(T v, int src, int width = 64) {
  return hipemu::exchange<T>(v, v, [src, width](int lane) { return (lane & ~(width - 1)) | (src & (width - 1)); });
}
template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
  return hipemu::exchange<T>(v, v, [mask](int lane) { return lane ^ mask; });
}
template <class T>
static inline T __shfl_down(T v, unsigned d, int width = 64) {
  return hipemu::exchange<T>(v, v, [d, width](int lane) { return (lane & (width - 1)) + int(d) < width ? lane + int(d) : lane; });
}
template <class T>
static inline T __shfl_up(T v, unsigned d, int width = 64) {
  return hipemu::exchange<T>(v, v, [d, width](int lane) { return (lane & (width - 1)) >= int(d) ? lane - int(d) : lane; });
}
static inline unsigned long long __ballot(int pred) {
  uint64_t* s = hipemu::wave_slots();
  s[hipemu::self()->lane] = pred ? 1 : 0;
  hipemu::wave_barrier();
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) m |= (unsigned long long)(s[l] & 1) << l;  // (dead lanes left their slot at 0)
  hipemu::wave_barrier();
  s[hipemu::self()->lane] = 0;
  return m;
}

// v_mfma_f32_16x16x4_f32: D = A (16x4) B (4x16) + C; lane l holds A[l & 15][l >> 4], B[l >> 4][l & 15] and
// C/D[(l >> 4) * 4 + r][l & 15], r = 0..3 (MI355X_MICROARCH / cdna_hip_programming guides)
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c) {
  uint64_t* s = hipemu::wave_slots();
  const int lane = hipemu::self()->lane;
  s[lane] = hipemu::bits(a);
  s[64 + lane] = hipemu::bits(b);
  hipemu::wave_barrier();
  const int col = lane & 15, rb = (lane >> 4) * 4;
  hipemu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    float t = c[r];
    for (int k = 0; k < 4; ++k)
      t = std::fmaf(hipemu::from_bits<float>(s[k * 16 + rb + r]), hipemu::from_bits<float>(s[64 + k * 16 + col]), t);
    d[r] = t;
  }
  hipemu::wave_barrier();
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu_mfma_f32_16x16x4f32(a, b, c)
// v_mfma_f64_16x16x4_f64: same operand layout; C/D[(l >> 4) + 4 r][l & 15], r = 0..3 (measured on gfx950:
// scripts/microbench/mfma_f64_layout.hip)
typedef double hipemu_f64x4 __attribute__((ext_vector_type(4)));
static inline hipemu_f64x4 hipemu_mfma_f64_16x16x4f64(double a, double b, hipemu_f64x4 c) {
  uint64_t* s = hipemu::wave_slots();
  const int lane = hipemu::self()->lane;
  s[lane] = hipemu::bits(a);
  s[64 + lane] = hipemu::bits(b);
  hipemu::wave_barrier();
  const int col = lane & 15, r0 = lane >> 4;
  hipemu_f64x4 d = c;
  for (int r = 0; r < 4; ++r) {
    double t = c[r];
    for (int k = 0; k < 4; ++k)
      t = std::fma(hipemu::from_bits<double>(s[k * 16 + r0 + 4 * r]), hipemu::from_bits<double>(s[64 + k * 16 + col]), t);
    d[r] = t;
  }
  hipemu::wave_barrier();
  return d;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) hipemu_mfma_f64_16x16x4f64(a, b, c)
