// Test double for librccl: the handful of entry points libprmf_hip.so binds with dlopen (PRMF_RCCL_LIB), implemented
// for ranks that are processes on ONE machine through POSIX shared memory, executed synchronously at the call
// (hipStreamSynchronize, device -> shared memory, barrier, reduce in rank order, barrier, -> device).  It lets the
// in-library multi-rank step (prmf_step_dist: which buffers, which counts, which streams, in-place all-reduce) run with
// two ranks on the single GPU of the test box, where the real RCCL refuses two ranks on one device.
//   g++ -O2 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include mock_rccl.cpp -o libmock_rccl.so \
//       -L/opt/rocm/lib -lamdhip64 -lrt -lpthread
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <vector>

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7,
               ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}

namespace {

constexpr size_t SLOT_BYTES = 32u << 20;
constexpr int MAX_RANKS = 8;
struct Shared {
  std::atomic<int> ready;
  pthread_barrier_t barrier;
  int nranks;
  alignas(64) unsigned char data[64];   // MAX_RANKS slots of SLOT_BYTES follow
};
struct Comm {
  Shared* sh;
  size_t bytes;
  int nranks, rank;
  char name[128];
};

size_t elem_size(ncclDataType_t t) { return t == ncclFloat64 ? 8 : (t == ncclFloat32 ? 4 : 0); }
unsigned char* slot(Comm* c, int r) { return c->sh->data + (size_t)r * SLOT_BYTES; }

// every rank deposits `bytes` from its device buffer, then all ranks see all slots
ncclResult_t deposit(Comm* c, const void* dev, size_t bytes, hipStream_t s) {
  if (bytes > SLOT_BYTES) return ncclInvalidArgument;
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(slot(c, c->rank), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  pthread_barrier_wait(&c->sh->barrier);
  return ncclSuccess;
}

template <typename T>
void sum_ranks(Comm* c, size_t first, size_t count, T* out) {
  for (size_t i = 0; i < count; ++i) {
    T acc = reinterpret_cast<const T*>(slot(c, 0))[first + i];
    for (int r = 1; r < c->nranks; ++r) acc += reinterpret_cast<const T*>(slot(c, r))[first + i];
    out[i] = acc;
  }
}

ncclResult_t reduce_to(Comm* c, size_t first, size_t count, ncclDataType_t t, void* recv_dev) {
  const size_t es = elem_size(t);
  std::vector<unsigned char> host(count * es);
  if (t == ncclFloat32) sum_ranks<float>(c, first, count, reinterpret_cast<float*>(host.data()));
  else sum_ranks<double>(c, first, count, reinterpret_cast<double*>(host.data()));
  pthread_barrier_wait(&c->sh->barrier);                      // everyone has read the slots: they may be overwritten
  if (hipMemcpy(recv_dev, host.data(), count * es, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

}  // namespace

extern "C" {

typedef Comm* ncclComm_t;

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id, 0, sizeof *id);
  std::snprintf(id->internal, sizeof id->internal, "/prmf_mock_rccl_%d_%ld", (int)getpid(), (long)time(nullptr));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm* c = new Comm();
  c->nranks = nranks;
  c->rank = rank;
  std::memcpy(c->name, id.internal, sizeof c->name);
  c->bytes = sizeof(Shared) + (size_t)MAX_RANKS * SLOT_BYTES;
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) return ncclSystemError;
  } else {
    for (int tries = 0; tries < 20000 && fd < 0; ++tries) {
      fd = shm_open(c->name, O_RDWR, 0600);
      if (fd < 0) usleep(1000);
    }
    if (fd < 0) return ncclSystemError;
    for (int tries = 0; tries < 20000; ++tries) {           // wait until rank 0 sized it
      if (lseek(fd, 0, SEEK_END) >= (off_t)c->bytes) break;
      usleep(1000);
    }
  }
  void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  c->sh = static_cast<Shared*>(p);
  if (rank == 0) {
    pthread_barrierattr_t a;
    pthread_barrierattr_init(&a);
    pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
    pthread_barrier_init(&c->sh->barrier, &a, (unsigned)nranks);
    c->sh->nranks = nranks;
    c->sh->ready.store(1, std::memory_order_release);
  } else {
    while (c->sh->ready.load(std::memory_order_acquire) != 1) usleep(1000);
  }
  pthread_barrier_wait(&c->sh->barrier);
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->sh, c->bytes);
  delete c;
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

// the mock executes every collective synchronously at the call, so a group is just its members in order
ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl error"; }

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
  if (op != ncclSum || !elem_size(t)) return ncclInvalidArgument;
  ncclResult_t r = deposit(c, send, count * elem_size(t), s);
  if (r != ncclSuccess) return r;
  return reduce_to(c, 0, count, t, recv);
}

ncclResult_t ncclReduceScatter(const void* send, void* recv, size_t recvcount, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c,
                               hipStream_t s) {
  if (op != ncclSum || !elem_size(t)) return ncclInvalidArgument;
  ncclResult_t r = deposit(c, send, recvcount * c->nranks * elem_size(t), s);
  if (r != ncclSuccess) return r;
  return reduce_to(c, recvcount * c->rank, recvcount, t, recv);
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t t, ncclComm_t c, hipStream_t s) {
  const size_t es = elem_size(t);
  if (!es) return ncclInvalidArgument;
  ncclResult_t r = deposit(c, send, sendcount * es, s);
  if (r != ncclSuccess) return r;
  std::vector<unsigned char> host(sendcount * es * c->nranks);
  for (int q = 0; q < c->nranks; ++q) std::memcpy(host.data() + (size_t)q * sendcount * es, slot(c, q), sendcount * es);
  pthread_barrier_wait(&c->sh->barrier);
  if (hipMemcpy(recv, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

}  // extern "C"
