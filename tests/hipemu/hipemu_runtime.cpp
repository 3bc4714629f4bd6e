// tests/hipemu/hipemu_runtime.cpp - scheduler and runtime API of the CPU execution harness (see hip/hip_runtime.h in this
// directory: TEST INFRASTRUCTURE ONLY).
//
// A kernel launch hands its workgroups to a small pool of OS threads (one workgroup at a time per thread). Inside a workgroup every work-item is
// a fiber (own stack, hand-written context switch); fibers run until they reach a synchronisation point:
//   block barrier  (__syncthreads)                     - all live work-items of the workgroup
//   wave barrier   (every wave-level operation)        - all live lanes of the wavefront (64 consecutive work-items)
// A work-item that returns from the kernel leaves both sets (like a lane whose EXEC bit is off for good). If no fiber can
// make progress the harness aborts with a diagnostic instead of hanging.
// Streams execute synchronously; a stream in capture mode records closures, a graph launch replays them.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <dlfcn.h>
#include <sched.h>

#include "hip/hip_runtime.h"

extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  subq $8, %rsp
  stmxcsr (%rsp)
  fnstcw 4(%rsp)
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  ldmxcsr (%rsp)
  fldcw 4(%rsp)
  addq $8, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

enum Wait { RUN = 0, WAIT_WAVE, WAIT_BLOCK, DONE };

struct Wave {
  int alive = 0, arrived = 0;
  uint64_t slots[7 * 64];
};
struct Fiber {
  void* sp = nullptr;
  Wait wait = RUN;
  Self self;
  Wave* wave = nullptr;
  void* site = nullptr;          // return address of the last barrier call (deadlock report)
  void* asan_fake = nullptr;     // AddressSanitizer builds (HIPEMU_SANITIZE=address): the fiber's fake-stack handle
  const void* stack_lo = nullptr;
};
struct Block {
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  int alive = 0, arrived = 0;
  const std::function<void()>* body = nullptr;
};

static thread_local Block* g_block = nullptr;
static thread_local Fiber* g_cur = nullptr;
static thread_local void* g_sched_sp = nullptr;
static thread_local std::vector<char>* g_stacks = nullptr;
constexpr size_t kStack = 96 * 1024;

Self* self() { return &g_cur->self; }
uint64_t* wave_slots() { return g_cur->wave->slots; }
int first_live_lane() {
  Block& b = *g_block;
  const int w = g_cur->self.wave;
  for (int l = 0; l < 64; ++l) {
    const size_t t = size_t(w) * 64 + l;
    if (t < b.fibers.size() && b.fibers[t].wait != DONE) return l;
  }
  return 0;
}

// AddressSanitizer has to be told about every stack switch (tests/hipemu/README.md: HIPEMU_SANITIZE=address)
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPEMU_ASAN 1
#endif
#endif
#ifdef HIPEMU_ASAN
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
extern "C" void __asan_unpoison_memory_region(void const volatile* addr, size_t size);
static thread_local const void* g_sched_stack_lo = nullptr;
static thread_local size_t g_sched_stack_size = 0;
static thread_local void* g_sched_fake = nullptr;
#endif

static void yield() {
#ifdef HIPEMU_ASAN
  Fiber* f = g_cur;
  __sanitizer_start_switch_fiber(f->wait == DONE ? nullptr : &f->asan_fake, g_sched_stack_lo, g_sched_stack_size);
  hipemu_switch(&f->sp, g_sched_sp);
  __sanitizer_finish_switch_fiber(f->asan_fake, &g_sched_stack_lo, &g_sched_stack_size);
#else
  hipemu_switch(&g_cur->sp, g_sched_sp);
#endif
}

static void release_wave(Block& b, Wave* w) {
  for (auto& f : b.fibers)
    if (f.wave == w && f.wait == WAIT_WAVE) f.wait = RUN;
  w->arrived = 0;
}
static void release_block(Block& b) {
  for (auto& f : b.fibers)
    if (f.wait == WAIT_BLOCK) f.wait = RUN;
  b.arrived = 0;
}

void wave_barrier() {
  Fiber* f = g_cur;
  f->site = __builtin_return_address(0);
  Wave* w = f->wave;
  if (w->alive <= 1) return;
  if (++w->arrived == w->alive) {
    release_wave(*g_block, w);
    return;
  }
  f->wait = WAIT_WAVE;
  yield();
}
// s_sleep inside a spin loop (persistent kernels that wait for another workgroup's data): the work-item gives way to
// the other work-items of its workgroup - the one it waits for may be among them - and, now and then, the OS thread to the
// threads that run the other workgroups
void spin_yield() {
  static thread_local unsigned n = 0;
  if ((++n & 0xfff) == 0) std::this_thread::yield();
  if (g_block->alive > 1) yield();  // (state stays RUN: the scheduler loop comes back to it)
}
void block_barrier() {
  Fiber* f = g_cur;
  f->site = __builtin_return_address(0);
  Block& b = *g_block;
  if (b.alive <= 1) return;
  if (++b.arrived == b.alive) {
    release_block(b);
    return;
  }
  f->wait = WAIT_BLOCK;
  yield();
}

static void trampoline() {
  Fiber* f = g_cur;
#ifdef HIPEMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &g_sched_stack_lo, &g_sched_stack_size);
#endif
  (*g_block->body)();
  // the work-item has returned: it leaves the wavefront and the workgroup
  Block& b = *g_block;
  f->wait = DONE;
  --b.alive;
  --f->wave->alive;
  if (f->wave->alive > 0 && f->wave->arrived == f->wave->alive) release_wave(b, f->wave);
  if (b.alive > 0 && b.arrived == b.alive) release_block(b);
  yield();
  std::fprintf(stderr, "hipemu: a finished fiber was resumed\n");
  std::abort();
}

static void prepare(Fiber& f, char* stack_top) {
  // layout expected by hipemu_switch when it loads this context: [mxcsr|fpucw][r15][r14][r13][r12][rbx][rbp][ret]
  uint64_t* sp = reinterpret_cast<uint64_t*>(reinterpret_cast<uintptr_t>(stack_top) & ~uintptr_t(15));
  *--sp = 0;                                        // fake return address of trampoline (keeps rsp = 8 mod 16 at entry)
  *--sp = reinterpret_cast<uint64_t>(&trampoline);  // ret target
  for (int i = 0; i < 6; ++i) *--sp = 0;            // rbp rbx r12 r13 r14 r15
  --sp;
  uint32_t* cw = reinterpret_cast<uint32_t*>(sp);
  cw[0] = 0x1f80;  // MXCSR default
  cw[1] = 0x037f;  // x87 control word default
  f.sp = sp;
}

static void run_block(const std::function<void()>& body, dim3 grid, dim3 block, uint3 bid) {
  const size_t n = size_t(block.x) * block.y * block.z;
  Block b;
  b.body = &body;
  b.fibers.resize(n);
  b.waves.resize((n + 63) / 64);
  b.alive = int(n);
  if (!g_stacks) g_stacks = new std::vector<char>();
  if (g_stacks->size() < n * kStack) g_stacks->resize(n * kStack);
#ifdef HIPEMU_ASAN
  // the frames a finished fiber was abandoned in (trampoline -> yield) are still poisoned: the top of every stack
  for (size_t t = 0; t < n; ++t) __asan_unpoison_memory_region(g_stacks->data() + (t + 1) * kStack - 8192, 8192);
#endif
  for (size_t t = 0; t < n; ++t) {
    Fiber& f = b.fibers[t];
    f.self.tid = uint3{unsigned(t % block.x), unsigned((t / block.x) % block.y), unsigned(t / (size_t(block.x) * block.y))};
    f.self.bid = bid;
    f.self.bdim = block;
    f.self.gdim = grid;
    f.self.lane = int(t & 63);
    f.self.wave = int(t >> 6);
    f.wave = &b.waves[t >> 6];
    ++f.wave->alive;
    std::memset(f.wave->slots, 0, sizeof f.wave->slots);
    prepare(f, g_stacks->data() + (t + 1) * kStack);
    f.stack_lo = g_stacks->data() + t * kStack;
  }
  g_block = &b;
  while (b.alive > 0) {
    bool progress = false;
    for (size_t t = 0; t < n; ++t) {
      Fiber& f = b.fibers[t];
      if (f.wait != RUN) continue;
      g_cur = &f;
#ifdef HIPEMU_ASAN
      __sanitizer_start_switch_fiber(&g_sched_fake, f.stack_lo, kStack);
      hipemu_switch(&g_sched_sp, f.sp);
      __sanitizer_finish_switch_fiber(g_sched_fake, nullptr, nullptr);
#else
      hipemu_switch(&g_sched_sp, f.sp);
#endif
      progress = true;
    }
    if (!progress) {
      int ww = 0, wb = 0;
      for (auto& f : b.fibers) {
        ww += f.wait == WAIT_WAVE;
        wb += f.wait == WAIT_BLOCK;
      }
      std::fprintf(stderr,
                   "hipemu: deadlock in block (%u,%u,%u): %d work-items alive, %d wait at a wave-level operation, %d at "
                   "__syncthreads (divergent synchronisation?)\n",
                   bid.x, bid.y, bid.z, b.alive, ww, wb);
      if (ww <= 64) {
        std::fprintf(stderr, "hipemu: work-items waiting at the wave-level operation:");
        for (size_t t = 0; t < n; ++t)
          if (b.fibers[t].wait == WAIT_WAVE) std::fprintf(stderr, " %zu", t);
        std::fprintf(stderr, "\n");
      }
      {
        // where they wait: return addresses into the library (addr2line -e <lib> <address - load base>)
        std::vector<std::pair<void*, int>> sites;
        for (auto& f : b.fibers) {
          if (f.wait != WAIT_WAVE && f.wait != WAIT_BLOCK) continue;
          bool found = false;
          for (auto& s : sites)
            if (s.first == f.site) {
              ++s.second;
              found = true;
            }
          if (!found) sites.push_back({f.site, 1});
        }
        Dl_info info;
        for (auto& s : sites) {
          const bool ok = dladdr(s.first, &info) != 0;
          std::fprintf(stderr, "hipemu:   %d work-items behind the call at %p (%s + 0x%zx)\n", s.second, s.first,
                       ok ? info.dli_fname : "?", ok ? size_t(static_cast<char*>(s.first) - static_cast<char*>(info.dli_fbase)) : size_t(0));
        }
      }
      std::abort();
    }
  }
  g_block = nullptr;
  g_cur = nullptr;
}

}  // namespace hipemu

// ---- streams, events, graphs ------------------------------------------------------------------------------------------
struct hipemu_graph {
  std::vector<std::function<void()>> nodes;
};
struct hipemu_stream {
  hipemu_graph* capturing = nullptr;
};
struct hipemu_event {
  std::chrono::steady_clock::time_point t;
};
static hipemu_stream g_null_stream;
static hipemu_stream* S(hipStream_t s) { return s ? s : &g_null_stream; }

namespace hipemu {
// workgroups of a grid are independent: a small pool of OS threads takes them from a shared counter
// (HIPEMU_THREADS, default: the CPUs this process may use, at most 16; 1 = everything on the calling thread)
struct Pool {
  std::vector<std::thread> workers;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  uint64_t generation = 0;
  int busy = 0;
  bool stop = false;
  // the current job
  const std::function<void()>* body = nullptr;
  dim3 grid, block;
  std::atomic<uint64_t> next{0};
  uint64_t total = 0;

  void work() {
    for (;;) {
      const uint64_t i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= total) return;
      const unsigned x = unsigned(i % grid.x), y = unsigned((i / grid.x) % grid.y), z = unsigned(i / (uint64_t(grid.x) * grid.y));
      run_block(*body, grid, block, uint3{x, y, z});
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv_job.wait(lk, [&] { return stop || generation != seen; });
        if (stop) return;
        seen = generation;
      }
      work();
      {
        std::lock_guard<std::mutex> lk(m);
        if (--busy == 0) cv_done.notify_all();
      }
    }
  }
  explicit Pool(int n) {
    for (int i = 0; i < n; ++i) workers.emplace_back([this] { loop(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv_job.notify_all();
    for (auto& w : workers) w.join();
  }
  void run(dim3 g, dim3 b, const std::function<void()>& f) {
    {
      std::lock_guard<std::mutex> lk(m);
      body = &f;
      grid = g;
      block = b;
      total = uint64_t(g.x) * g.y * g.z;
      next.store(0);
      busy = int(workers.size());
      ++generation;
    }
    cv_job.notify_all();
    work();  // the calling thread takes its share
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return busy == 0; });
  }
};
static int pool_threads() {
  if (const char* e = std::getenv("HIPEMU_THREADS")) return std::max(1, std::atoi(e));
  cpu_set_t set;
  int n = 1;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
  return std::max(1, std::min(16, n));
}
static void run_grid(dim3 grid, dim3 block, const std::function<void()>& body) {
  // (one grid at a time: several HOST threads may launch - rba_create_sharded runs a thread per device)
  static std::mutex launch_mutex;
  std::lock_guard<std::mutex> launch_lock(launch_mutex);
  static const int threads = pool_threads();
  const uint64_t total = uint64_t(grid.x) * grid.y * grid.z;
  if (threads == 1 || total == 1) {
    for (unsigned z = 0; z < grid.z; ++z)
      for (unsigned y = 0; y < grid.y; ++y)
        for (unsigned x = 0; x < grid.x; ++x) run_block(body, grid, block, uint3{x, y, z});
    return;
  }
  static Pool pool(threads - 1);
  pool.run(grid, block, body);
}
void launch(dim3 grid, dim3 block, size_t /*shmem*/, hipStream_t stream, std::function<void()> body) {
  if (S(stream)->capturing) {
    S(stream)->capturing->nodes.push_back([grid, block, body]() { run_grid(grid, block, body); });
    return;
  }
  run_grid(grid, block, body);
}
}  // namespace hipemu

hipError_t hipGetDeviceCount(int* n) {
  *n = 8;  // "one node": every rank of a multi-process run finds its device index
  return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) {
  *d = 0;
  return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
  if (a == hipDeviceAttributeWallClockRate) {  // (wall_clock64() is the time-stamp counter here: a nominal 3 GHz, in kHz)
    *v = 3000000;
    return hipSuccess;
  }
  // "compute units": keeps the persistent kernels' grids small. A kernel whose workgroups wait for each other
  // (kernels_pcgp.hpp) needs them all running at once: HIPEMU_CUS <= HIPEMU_THREADS
  static const int cus = [] {
    const char* e = std::getenv("HIPEMU_CUS");
    return e ? std::max(1, std::atoi(e)) : 4;
  }();
  *v = cus;
  return hipSuccess;
}
// Stream capture (the product captures with hipStreamCaptureModeThreadLocal): while THIS thread captures, the calls the real
// runtime refuses - allocation, synchronous copies / memsets, device-wide and capturing-stream synchronisation - fail here
// too, with a message, instead of silently working (which they would: everything is synchronous on the harness)
static thread_local int t_captures = 0;
static bool refused_during_capture(const char* what) {
  if (t_captures == 0) return false;
  std::fprintf(stderr, "hipemu: %s while this thread captures a stream (hipErrorStreamCaptureUnsupported)\n", what);
  return true;
}
// HIPEMU_POISON=1: fresh device memory is filled with 0xff (NaN as float / double, -1 as int) instead of whatever the host
// allocator returns (zero pages for large buffers): a kernel that consumes memory nobody wrote shows up in the results
static bool poison_fresh_memory() {
  static const bool on = [] { const char* e = std::getenv("HIPEMU_POISON"); return e && *e && *e != '0'; }();
  return on;
}
#ifdef HIPEMU_ASAN
// AddressSanitizer builds: no slack behind a buffer - the bytes between the requested size and the aligned size are
// poisoned, so that an access one element past the end of a device buffer is reported
extern "C" void __asan_poison_memory_region(void const volatile* addr, size_t size);
static std::mutex g_alloc_mutex;
static std::unordered_map<void*, std::pair<size_t, size_t>> g_alloc_sizes;
hipError_t hipMalloc(void** p, size_t n) {
  if (refused_during_capture("hipMalloc")) return hipErrorStreamCaptureUnsupported;
  const size_t rounded = (std::max<size_t>(n, 1) + 255) / 256 * 256;
  *p = std::aligned_alloc(256, rounded);
  if (!*p) return hipErrorInvalidValue;
  if (poison_fresh_memory()) std::memset(*p, 0xff, n);
  if (rounded > n) __asan_poison_memory_region(static_cast<char*>(*p) + n, rounded - n);
  std::lock_guard<std::mutex> lk(g_alloc_mutex);
  g_alloc_sizes[*p] = {n, rounded};
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  if (refused_during_capture("hipFree")) return hipErrorStreamCaptureUnsupported;
  if (p) {
    std::lock_guard<std::mutex> lk(g_alloc_mutex);
    auto it = g_alloc_sizes.find(p);
    if (it != g_alloc_sizes.end()) {
      hipemu::__asan_unpoison_memory_region(static_cast<char*>(p) + it->second.first, it->second.second - it->second.first);
      g_alloc_sizes.erase(it);
    }
  }
  std::free(p);
  return hipSuccess;
}
#else
hipError_t hipMalloc(void** p, size_t n) {
  if (refused_during_capture("hipMalloc")) return hipErrorStreamCaptureUnsupported;
  *p = std::aligned_alloc(256, (n + 255) / 256 * 256 + 256);
  if (*p && poison_fresh_memory()) std::memset(*p, 0xff, (n + 255) / 256 * 256 + 256);
  return *p ? hipSuccess : hipErrorInvalidValue;
}
hipError_t hipFree(void* p) {
  if (refused_during_capture("hipFree")) return hipErrorStreamCaptureUnsupported;
  std::free(p);
  return hipSuccess;
}
#endif
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind k, hipStream_t s) {
  if (S(s)->capturing) {
    // (the source of a captured copy is read at replay time, as in a real graph; host staging buffers of the
    //  callers outlive the graph)
    S(s)->capturing->nodes.push_back([dst, src, n]() { std::memmove(dst, src, n); });
    return hipSuccess;
  }
  (void)k;
  std::memmove(dst, src, n);
  return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) {
  if (refused_during_capture("hipMemcpy")) return hipErrorStreamCaptureUnsupported;
  std::memmove(dst, src, n);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t s) {
  if (S(s)->capturing) {
    S(s)->capturing->nodes.push_back([dst, v, n]() { std::memset(dst, v, n); });
    return hipSuccess;
  }
  std::memset(dst, v, n);
  return hipSuccess;
}
hipError_t hipMemset(void* dst, int v, size_t n) {
  if (refused_during_capture("hipMemset")) return hipErrorStreamCaptureUnsupported;
  std::memset(dst, v, n);
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  *s = new hipemu_stream();
  return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t* s) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipStreamDestroy(hipStream_t s) {
  delete s;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
  if (S(s)->capturing && refused_during_capture("hipStreamSynchronize on the capturing stream")) return hipErrorStreamCaptureUnsupported;
  return hipSuccess;
}
hipError_t hipStreamQuery(hipStream_t s) {
  if (S(s)->capturing && refused_during_capture("hipStreamQuery on the capturing stream")) return hipErrorStreamCaptureUnsupported;
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize() {
  if (refused_during_capture("hipDeviceSynchronize")) return hipErrorStreamCaptureUnsupported;
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) {
  *e = new hipemu_event();
  (*e)->t = std::chrono::steady_clock::now();
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
  delete e;
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  if (S(s)->capturing) {
    S(s)->capturing->nodes.push_back([e]() { e->t = std::chrono::steady_clock::now(); });
    return hipSuccess;
  }
  e->t = std::chrono::steady_clock::now();
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
// (two workgroups per CU: persistent kernels walk several tiles per wave on the harness as well)
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) {
  *n = 2;
  return hipSuccess;
}  // streams execute at enqueue: a recorded event has completed
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) {
  S(s)->capturing = new hipemu_graph();
  ++t_captures;
  return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g) {
  *g = S(s)->capturing;
  S(s)->capturing = nullptr;
  if (t_captures > 0) --t_captures;
  return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t) {
  *e = new hipemu_graph(*g);
  return hipSuccess;
}
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s) {
  if (S(s)->capturing) {  // (a graph launched into a capturing stream becomes part of it)
    for (auto& n : e->nodes) S(s)->capturing->nodes.push_back(n);
    return hipSuccess;
  }
  for (auto& n : e->nodes) n();
  return hipSuccess;
}
hipError_t hipGraphExecDestroy(hipGraphExec_t e) {
  delete e;
  return hipSuccess;
}
hipError_t hipGraphDestroy(hipGraph_t g) {
  delete g;
  return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }

// ---- dynamic LDS of the product's kernels (`extern __shared__` declarations, rewritten to `extern` by build_emu.py) ----
namespace rba {
alignas(16) thread_local char smem_raw[160 * 1024];
alignas(16) thread_local unsigned char hx_lds_raw[160 * 1024];
alignas(16) thread_local char smem_pcgs[160 * 1024];
alignas(16) thread_local char smem_s1[160 * 1024];
alignas(16) thread_local char smem_s1c[160 * 1024];
alignas(16) thread_local char smem_a64[160 * 1024];
alignas(16) thread_local char smem_pg[160 * 1024];
}  // namespace rba
