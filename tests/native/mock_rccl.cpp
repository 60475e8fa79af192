// Test double for librccl: the entry points libprmf_hip.so binds with dlopen (PRMF_RCCL_LIB), implemented for ranks
// that live on ONE machine -- processes, threads of one process, or a mix -- through POSIX shared memory, executed
// synchronously at the call (hipStreamSynchronize, device -> shared memory, barrier, reduce in rank order, barrier,
// -> device).  It lets the in-library multi-rank step (prmf_step_dist: which buffers, which counts, which streams,
// in-place all-reduce) run with up to 8 ranks on the single GPU of the test box, where the real RCCL refuses two ranks
// on one device.  Nothing here is process-global per rank: a rank is its ncclComm_t, so 8 ranks can be 8 threads (the
// GPU box allows few processes on its card).
//
// Failure model (tests/test_gpu_eightway.py::test_failed_collective_*):
//   MOCK_RCCL_FAIL="<rank>:<n>"   the n-th collective (0-based, counted per communicator) of that rank returns
//                                 ncclInternalError WITHOUT taking part -- its peers are left inside the collective
//   ncclCommAbort                 raises the shared abort flag: every rank waiting in (or later entering) a collective of
//                                 this communicator returns ncclRemoteError instead of waiting for ever
//   MOCK_RCCL_TIMEOUT_S           (default 120) a rank that waits longer than this for its peers raises the abort flag
//                                 itself and returns ncclSystemError
// Build: g++ -O2 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include mock_rccl.cpp -o libmock_rccl.so
//            -L/opt/rocm/lib -lamdhip64 -lrt -lpthread
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7,
               ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}

namespace {

constexpr size_t SLOT_BYTES = 32u << 20;
constexpr int MAX_RANKS = 8;
struct Shared {
  std::atomic<int> ready;
  std::atomic<int> abort;
  std::atomic<unsigned> arrived, generation, detached;
  int nranks;
  alignas(64) unsigned char data[64];   // MAX_RANKS slots of SLOT_BYTES follow
};
struct Comm {
  Shared* sh;
  size_t bytes;
  int nranks, rank;
  long calls;            // collectives issued on this communicator so far
  long fail_at;          // MOCK_RCCL_FAIL: index of the collective that fails on this rank, or -1
  double timeout_s;
  char name[128];
};

size_t elem_size(ncclDataType_t t) { return t == ncclFloat64 ? 8 : (t == ncclFloat32 ? 4 : 0); }
unsigned char* slot(Comm* c, int r) { return c->sh->data + (size_t)r * SLOT_BYTES; }

// counting barrier in the shared segment; gives up when the communicator was aborted or the peers do not arrive
ncclResult_t barrier(Comm* c) {
  Shared* sh = c->sh;
  if (sh->abort.load(std::memory_order_acquire)) return ncclRemoteError;
  const unsigned gen = sh->generation.load(std::memory_order_acquire);
  if (sh->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned)c->nranks) {
    sh->arrived.store(0, std::memory_order_relaxed);
    sh->generation.fetch_add(1, std::memory_order_acq_rel);
    return ncclSuccess;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0; sh->generation.load(std::memory_order_acquire) == gen; ++spins) {
    if (sh->abort.load(std::memory_order_acquire)) return ncclRemoteError;
    if (spins < 2000) {
      sched_yield();
    } else {
      usleep(50);
      if ((spins & 1023) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) {
        sh->abort.store(1, std::memory_order_release);
        return ncclSystemError;
      }
    }
  }
  return ncclSuccess;
}

// the failure injection: does this call fail on this rank?
bool injected_failure(Comm* c) { return c->fail_at >= 0 && c->calls++ == c->fail_at; }

// every rank deposits `bytes` from its device buffer, then all ranks see all slots
ncclResult_t deposit(Comm* c, const void* dev, size_t bytes, hipStream_t s) {
  if (bytes > SLOT_BYTES) return ncclInvalidArgument;
  // every copy runs ON THE CALLER'S STREAM and is waited for there: a plain hipMemcpy uses the null stream, which a
  // non-blocking stream (torch's, one per rank thread) is not ordered with -- and for pageable host memory it may return
  // before the DMA has reached the device, so the next kernel on `s` could read a half-written receive buffer
  if (bytes && hipMemcpyAsync(slot(c, c->rank), dev, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  return barrier(c);
}

template <typename T>
void sum_ranks(Comm* c, size_t first, size_t count, T* out) {
  for (size_t i = 0; i < count; ++i) {
    T acc = reinterpret_cast<const T*>(slot(c, 0))[first + i];
    for (int r = 1; r < c->nranks; ++r) acc += reinterpret_cast<const T*>(slot(c, r))[first + i];
    out[i] = acc;
  }
}

ncclResult_t to_device(void* dev, const void* host, size_t bytes, hipStream_t s) {
  if (!bytes) return ncclSuccess;
  if (hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
  return hipStreamSynchronize(s) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;      // `host` is a local buffer
}

ncclResult_t reduce_to(Comm* c, size_t first, size_t count, ncclDataType_t t, void* recv_dev, hipStream_t s) {
  const size_t es = elem_size(t);
  std::vector<unsigned char> host(count * es);
  if (t == ncclFloat32) sum_ranks<float>(c, first, count, reinterpret_cast<float*>(host.data()));
  else sum_ranks<double>(c, first, count, reinterpret_cast<double*>(host.data()));
  const ncclResult_t r = barrier(c);                          // everyone has read the slots: they may be overwritten
  if (r != ncclSuccess) return r;
  return to_device(recv_dev, host.data(), count * es, s);
}

void detach(Comm* c) {
  if (!c) return;
  // the last rank to leave removes the name (rank 0 may be gone long before its peers have even attached otherwise)
  if (c->sh->detached.fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned)c->nranks) shm_unlink(c->name);
  munmap(c->sh, c->bytes);
  delete c;
}

}  // namespace

extern "C" {

typedef Comm* ncclComm_t;

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  static std::atomic<unsigned> serial{0};
  std::memset(id, 0, sizeof *id);
  std::snprintf(id->internal, sizeof id->internal, "/prmf_mock_rccl_%d_%ld_%u", (int)getpid(), (long)time(nullptr), serial.fetch_add(1));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm* c = new Comm();
  c->nranks = nranks;
  c->rank = rank;
  c->calls = 0;
  c->fail_at = -1;
  c->timeout_s = 120.0;
  if (const char* e = getenv("MOCK_RCCL_TIMEOUT_S")) c->timeout_s = atof(e);
  if (const char* e = getenv("MOCK_RCCL_FAIL")) {
    int fr = -1;
    long fn = -1;
    if (std::sscanf(e, "%d:%ld", &fr, &fn) == 2 && fr == rank) c->fail_at = fn;
  }
  std::memcpy(c->name, id.internal, sizeof c->name);
  c->bytes = sizeof(Shared) + (size_t)MAX_RANKS * SLOT_BYTES;
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { delete c; return ncclSystemError; }
  } else {
    for (int tries = 0; tries < 20000 && fd < 0; ++tries) {
      fd = shm_open(c->name, O_RDWR, 0600);
      if (fd < 0) usleep(1000);
    }
    if (fd < 0) { delete c; return ncclSystemError; }
    for (int tries = 0; tries < 20000; ++tries) {           // wait until rank 0 sized it
      if (lseek(fd, 0, SEEK_END) >= (off_t)c->bytes) break;
      usleep(1000);
    }
  }
  void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return ncclSystemError; }
  c->sh = static_cast<Shared*>(p);
  if (rank == 0) {
    c->sh->nranks = nranks;                                  // a fresh segment is zero-filled: counters start at 0
    c->sh->ready.store(1, std::memory_order_release);
  } else {
    while (c->sh->ready.load(std::memory_order_acquire) != 1) usleep(1000);
  }
  const ncclResult_t r = barrier(c);
  if (r != ncclSuccess) { detach(c); return r; }
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  detach(c);
  return ncclSuccess;
}

// what a rank does with a communicator one of whose collectives failed: nobody may wait for this rank any more
ncclResult_t ncclCommAbort(ncclComm_t c) {
  if (c) c->sh->abort.store(1, std::memory_order_release);
  detach(c);
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int* count) {
  if (!c || !count) return ncclInvalidArgument;
  *count = c->nranks;
  return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t c, int* rank) {
  if (!c || !rank) return ncclInvalidArgument;
  *rank = c->rank;
  return ncclSuccess;
}

// the mock executes every collective synchronously at the call, so a group is just its members in order; the depth is
// kept per thread so that a caller that leaves a group open after an error (the bug VERDICT r3 named) is caught
thread_local int g_group_depth = 0;
ncclResult_t ncclGroupStart() { ++g_group_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  if (g_group_depth <= 0) return ncclInvalidUsage;
  --g_group_depth;
  return ncclSuccess;
}
int mock_rccl_group_depth() { return g_group_depth; }

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclRemoteError: return "mock rccl: the communicator was aborted by a peer";
    case ncclSystemError: return "mock rccl: system error / peers did not arrive in time";
    case ncclInternalError: return "mock rccl: injected failure";
    default: return "mock rccl error";
  }
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
  if (op != ncclSum || !elem_size(t)) return ncclInvalidArgument;
  if (injected_failure(c)) return ncclInternalError;
  ncclResult_t r = deposit(c, send, count * elem_size(t), s);
  if (r != ncclSuccess) return r;
  return reduce_to(c, 0, count, t, recv, s);
}

ncclResult_t ncclReduceScatter(const void* send, void* recv, size_t recvcount, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c,
                               hipStream_t s) {
  if (op != ncclSum || !elem_size(t)) return ncclInvalidArgument;
  if (injected_failure(c)) return ncclInternalError;
  ncclResult_t r = deposit(c, send, recvcount * c->nranks * elem_size(t), s);
  if (r != ncclSuccess) return r;
  return reduce_to(c, recvcount * c->rank, recvcount, t, recv, s);
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t t, ncclComm_t c, hipStream_t s) {
  const size_t es = elem_size(t);
  if (!es) return ncclInvalidArgument;
  if (injected_failure(c)) return ncclInternalError;
  ncclResult_t r = deposit(c, send, sendcount * es, s);
  if (r != ncclSuccess) return r;
  std::vector<unsigned char> host(sendcount * es * c->nranks);
  for (int q = 0; q < c->nranks; ++q) std::memcpy(host.data() + (size_t)q * sendcount * es, slot(c, q), sendcount * es);
  r = barrier(c);
  if (r != ncclSuccess) return r;
  return to_device(recv, host.data(), host.size(), s);
}

}  // extern "C"
