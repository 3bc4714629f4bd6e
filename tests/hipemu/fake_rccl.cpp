// tests/hipemu/fake_rccl.cpp - a stand-in for librccl.so.1 for runs on the CPU execution harness (TEST / DEVELOPMENT
// TOOL ONLY, see hip/hip_runtime.h in this directory): the entry points the library dlopens, implemented over files
// in a per-communicator directory so that several PROCESSES on this machine form a communicator without any GPU or
// network stack. It lets the multi-rank code path of the library (rba_comm_init, the union of the block structure over
// the ranks, every all-reduce site) and the N > 1 flow of bench.py run where there is no GPU. The sum is formed in rank
// order on every rank (bit-identical results on all ranks, like a ring all-reduce); nothing here says anything about the
// speed or the behaviour of RCCL itself.
// Build: build_emu.py -> _build/fake_rccl/librccl.so.1; the harness build of the library opens the path in HIPEMU_RCCL
// instead of "librccl.so.1" (a process that imported torch already holds torch's RCCL under that name).
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {
struct UniqueId {
  char internal[128];
};
struct Comm {
  std::string dir;
  int rank = 0, nranks = 1;
  uint64_t seq = 0;
};
std::string path(const Comm& c, uint64_t seq, int rank) {
  return c.dir + "/" + std::to_string(seq) + "." + std::to_string(rank);
}
bool exists(const std::string& p) {
  struct stat st;
  return ::stat(p.c_str(), &st) == 0;
}
size_t elem_size(int dtype) { return dtype == 8 ? 8 : 4; }  // ncclFloat64 = 8; ncclInt32 = 2, ncclFloat32 = 7
template <class T>
void reduce(T* acc, const T* v, size_t n, int op) {
  for (size_t i = 0; i < n; ++i) acc[i] = op == 2 ? (v[i] > acc[i] ? v[i] : acc[i]) : acc[i] + v[i];  // ncclMax = 2, ncclSum = 0
}
}  // namespace

extern "C" {
int ncclGetUniqueId(UniqueId* id) {
  std::memset(id, 0, sizeof *id);
  const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
  std::snprintf(id->internal, sizeof id->internal, "hipemu_rccl_%d_%lld", int(::getpid()), static_cast<long long>(now));
  return 0;
}
int ncclCommInitRank(void** comm, int nranks, UniqueId id, int rank) {
  auto* c = new Comm();
  const char* base = std::getenv("HIPEMU_RCCL_DIR");
  c->dir = std::string(base ? base : "/tmp") + "/" + std::string(id.internal, strnlen(id.internal, sizeof id.internal));
  c->rank = rank;
  c->nranks = nranks;
  ::mkdir(c->dir.c_str(), 0700);
  *comm = c;
  return 0;
}
int ncclCommDestroy(void* comm) {
  auto* c = static_cast<Comm*>(comm);
  for (uint64_t s = c->seq > 2 ? c->seq - 2 : 0; s <= c->seq; ++s) ::unlink(path(*c, s, c->rank).c_str());
  ::rmdir(c->dir.c_str());  // (succeeds for the last rank to leave)
  delete c;
  return 0;
}
int ncclCommCount(void* comm, int* n) {
  *n = static_cast<Comm*>(comm)->nranks;
  return 0;
}
const char* ncclGetErrorString(int) { return "fake_rccl error"; }
int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void* /*stream*/) {
  auto* c = static_cast<Comm*>(comm);
  const uint64_t seq = ++c->seq;
  const size_t bytes = count * elem_size(dtype);
  if (seq > 2) ::unlink(path(*c, seq - 2, c->rank).c_str());  // everybody is past it (they published seq - 1)
  {
    const std::string tmp = path(*c, seq, c->rank) + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return 1;
    std::fwrite(send, 1, bytes, f);
    std::fclose(f);
    if (std::rename(tmp.c_str(), path(*c, seq, c->rank).c_str()) != 0) return 1;
  }
  std::vector<char> acc(bytes), buf(bytes);
  for (int r = 0; r < c->nranks; ++r) {
    const std::string p = path(*c, seq, r);
    const auto t0 = std::chrono::steady_clock::now();
    while (!exists(p)) {
      std::this_thread::sleep_for(std::chrono::microseconds(100));
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(600)) return 2;  // a rank is gone
    }
    FILE* f = std::fopen(p.c_str(), "rb");
    if (!f || std::fread(buf.data(), 1, bytes, f) != bytes) return 3;
    std::fclose(f);
    if (r == 0) {
      acc = buf;
    } else if (dtype == 8) {
      reduce(reinterpret_cast<double*>(acc.data()), reinterpret_cast<const double*>(buf.data()), count, op);
    } else if (dtype == 7) {
      reduce(reinterpret_cast<float*>(acc.data()), reinterpret_cast<const float*>(buf.data()), count, op);
    } else {
      reduce(reinterpret_cast<int32_t*>(acc.data()), reinterpret_cast<const int32_t*>(buf.data()), count, op);
    }
  }
  std::memcpy(recv, acc.data(), bytes);
  return 0;
}
// ncclReduce: the same exchange, the sum lands on `root` only (in place elsewhere: the buffer keeps the rank's own data)
int ncclReduce(const void* send, void* recv, size_t count, int dtype, int op, int root, void* comm, void* stream) {
  auto* c = static_cast<Comm*>(comm);
  std::vector<char> out(count * elem_size(dtype));
  const int rc = ncclAllReduce(send, out.data(), count, dtype, op, comm, stream);
  if (rc != 0) return rc;
  if (c->rank == root) std::memcpy(recv, out.data(), out.size());
  return 0;
}
// (the calls of a group execute at once, in program order - the same order on every rank)
int ncclGroupStart() { return 0; }
int ncclGroupEnd() { return 0; }
}
