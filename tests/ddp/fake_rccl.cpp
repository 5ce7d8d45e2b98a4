// TEST INFRASTRUCTURE — a stand-in for librccl.so that lets SEVERAL ranks share ONE GPU (or no GPU at all for the bootstrap).
//
// Why it exists: dan_amd/csrc/comm.cpp issues the data-parallel gradient exchange (tf_replicate_model_fn.py:297-343, 633-645: add_n over the
// towers' gradients = an all-reduce over the ranks) straight on RCCL's C entry points.  Real RCCL refuses two ranks on one device and a GPU
// test box has exactly one, so the multi-rank arithmetic above those entry points — unique-id broadcast, ncclCommInitRank(N > 1), the
// reduce-scatter / all-gather shard offsets with world > 1, the bf16 wire form, the hipGraph-captured step with real peers, the async-error
// poll and the bootstrap deadline — could never run before the driver's 8-GPU node.  This file implements the thirteen nccl* symbols comm.cpp
// binds (eleven mandatory, two optional) with the SAME stream semantics: every collective is asynchronous work on the caller's stream
//     hipMemcpyAsync(device -> pinned host)  ->  hipLaunchHostFunc(exchange with the peers through POSIX shared memory)  ->
//     hipMemcpyAsync(pinned host -> device)
// so event ordering, side streams and stream capture behave as with the real library (memcpy and host nodes are capturable).
// Duplicate devices are accepted.  Nothing in dan_amd/ knows this file: tests pass its path to danhip_comm_load(path) (DANHIP_RCCL_PATH).
//
// Environment: FAKE_RCCL_TIMEOUT_S (default 120): deadline of every inter-rank wait; on expiry the communicator's async error becomes
// ncclSystemError and every later wait returns at once (ncclCommGetAsyncError reports it).  FAKE_RCCL_INIT_HANG=1: ncclCommInitRank waits
// for its peers for ever, as the real library does — the behaviour dan_amd.trainer's bootstrap deadline has to survive.
// FAKE_RCCL_SLOT_MB (default 8): bytes of one rank's exchange slot; larger collectives run in pieces.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
enum { kOk = 0, kUnhandledCuda = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5 };
enum { kHalf = 6, kFloat = 7, kBf16 = 9 };
enum { kAllReduce = 0, kReduceScatter = 1, kAllGather = 2 };

struct Header {                       // start of the shared segment; a fresh segment is all zeros
  std::atomic<uint32_t> count;        // central sense-reversing barrier
  std::atomic<uint32_t> generation;
  std::atomic<uint32_t> aborted;      // any rank gave up: nobody waits any more
  std::atomic<uint32_t> attached;
};
constexpr size_t kHeaderBytes = 4096;

double now_s() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

double env_seconds(const char* name, double dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atof(v) : dflt;
}

struct Comm;
struct Task {                         // one piece of one collective; owned by the communicator (a captured graph replays it)
  Comm* comm;
  int op, dtype;
  size_t n;                           // elements per rank in this piece
};

struct Comm {
  int nranks = 0, rank = 0, device = -1;
  char name[64] = {0};
  void* seg = nullptr;
  size_t seg_bytes = 0, slot_bytes = 0;
  Header* hdr = nullptr;
  char* stage_in = nullptr;           // pinned host: what this rank contributes to the current piece
  char* stage_out = nullptr;          // pinned host: what this rank receives
  std::atomic<int> async_error{kOk};
  double timeout_s = 120.0;
  std::vector<Task*> tasks;

  char* slot(int r) const { return (char*)seg + kHeaderBytes + (size_t)r * slot_bytes; }

  bool barrier(double timeout) {      // false: deadline passed or a peer aborted
    if (nranks == 1) return true;
    if (hdr->aborted.load()) return false;
    const uint32_t gen = hdr->generation.load();
    if (hdr->count.fetch_add(1) + 1 == (uint32_t)nranks) {
      hdr->count.store(0);
      hdr->generation.fetch_add(1);
      return true;
    }
    const double t0 = now_s();
    unsigned spins = 0;
    while (hdr->generation.load() == gen) {
      if (hdr->aborted.load()) return false;
      if ((++spins & 1023u) == 0) {
        if (timeout > 0 && now_s() - t0 > timeout) {
          hdr->aborted.store(1);
          return false;
        }
        usleep(50);
      } else {
        sched_yield();
      }
    }
    return true;
  }
};

size_t elem_size(int dtype) { return dtype == kFloat ? 4 : 2; }

float load_as_float(const char* p, size_t i, int dtype) {
  if (dtype == kFloat) return ((const float*)p)[i];
  const uint16_t h = ((const uint16_t*)p)[i];
  if (dtype == kBf16) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
  }
  return (float)(*(const _Float16*)&h);
}

void store_from_float(char* p, size_t i, int dtype, float v) {
  if (dtype == kFloat) {
    ((float*)p)[i] = v;
  } else if (dtype == kBf16) {        // round to nearest even, NaN kept
    uint32_t u;
    memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) {
      ((uint16_t*)p)[i] = (uint16_t)((u >> 16) | 0x40);
    } else {
      u += 0x7fffu + ((u >> 16) & 1u);
      ((uint16_t*)p)[i] = (uint16_t)(u >> 16);
    }
  } else {
    const _Float16 h = (_Float16)v;
    memcpy((uint16_t*)p + i, &h, 2);
  }
}

// Runs on the HIP runtime's callback thread, in stream order: no HIP call in here.
void exchange(void* arg) {
  Task* t = (Task*)arg;
  Comm* c = t->comm;
  const size_t es = elem_size(t->dtype);
  const size_t mine = (t->op == kReduceScatter ? t->n * c->nranks : t->n) * es;
  memcpy(c->slot(c->rank), c->stage_in, mine);
  if (!c->barrier(c->timeout_s)) {                       // every rank's contribution is in its slot
    c->async_error.store(kSystemError);
    return;
  }
  if (t->op == kAllGather) {
    for (int r = 0; r < c->nranks; ++r) memcpy(c->stage_out + (size_t)r * t->n * es, c->slot(r), t->n * es);
  } else {                                               // sums in rank order, fp32 accumulate, one rounding: every rank gets the same bits
    const size_t off = (t->op == kReduceScatter) ? (size_t)c->rank * t->n : 0;
    for (size_t i = 0; i < t->n; ++i) {
      float s = 0.f;
      for (int r = 0; r < c->nranks; ++r) s += load_as_float(c->slot(r), off + i, t->dtype);
      store_from_float(c->stage_out, i, t->dtype, s);
    }
  }
  if (!c->barrier(c->timeout_s)) c->async_error.store(kSystemError);   // every rank has read the slots: the next piece may overwrite them
}

int lazy_buffers(Comm* c) {
  if (c->stage_in) return kOk;
  if (hipHostMalloc((void**)&c->stage_in, c->slot_bytes, hipHostMallocDefault) != hipSuccess) return kUnhandledCuda;
  if (hipHostMalloc((void**)&c->stage_out, c->slot_bytes, hipHostMallocDefault) != hipSuccess) return kUnhandledCuda;
  return kOk;
}

#define FR_HIP(x)                          \
  do {                                     \
    if ((x) != hipSuccess) {               \
      c->async_error.store(kUnhandledCuda); \
      return kUnhandledCuda;               \
    }                                      \
  } while (0)

int collective(Comm* c, int op, const void* send, void* recv, size_t count, int dtype, hipStream_t stream) {
  if (!c || !c->seg) return kInvalidArgument;
  if (dtype != kFloat && dtype != kHalf && dtype != kBf16) return kInvalidArgument;
  if (c->async_error.load() != kOk) return c->async_error.load();
  if (int rc = lazy_buffers(c)) return rc;
  const size_t es = elem_size(dtype);
  // per-rank elements of one piece: the reduce-scatter contribution and the all-gather result hold nranks of them
  size_t piece = c->slot_bytes / es / (op == kAllReduce ? 1 : (size_t)c->nranks);
  piece &= ~(size_t)63;
  for (size_t o = 0; o < count; o += piece) {
    const size_t n = (count - o < piece) ? count - o : piece;
    Task* t = new Task{c, op, dtype, n};
    c->tasks.push_back(t);
    if (op == kReduceScatter) {
      for (int d = 0; d < c->nranks; ++d)
        FR_HIP(hipMemcpyAsync(c->stage_in + (size_t)d * n * es, (const char*)send + ((size_t)d * count + o) * es, n * es, hipMemcpyDeviceToHost, stream));
    } else {
      FR_HIP(hipMemcpyAsync(c->stage_in, (const char*)send + o * es, n * es, hipMemcpyDeviceToHost, stream));
    }
    FR_HIP(hipLaunchHostFunc(stream, exchange, t));
    if (op == kAllGather) {
      for (int r = 0; r < c->nranks; ++r)
        FR_HIP(hipMemcpyAsync((char*)recv + ((size_t)r * count + o) * es, c->stage_out + (size_t)r * n * es, n * es, hipMemcpyHostToDevice, stream));
    } else {
      FR_HIP(hipMemcpyAsync((char*)recv + o * es, c->stage_out, n * es, hipMemcpyHostToDevice, stream));
    }
  }
  return kOk;
}

void release(Comm* c) {
  if (c->stage_in) (void)hipHostFree(c->stage_in);
  if (c->stage_out) (void)hipHostFree(c->stage_out);
  if (c->seg) munmap(c->seg, c->seg_bytes);
  for (Task* t : c->tasks) delete t;
  delete c;
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef void* ncclComm_t;

int ncclGetVersion(int* v) {
  if (!v) return kInvalidArgument;
  *v = 29999;                                            // no real RCCL carries this number: a line that reports it ran on the stand-in
  return kOk;
}

const char* ncclGetErrorString(int e) {
  switch (e) {
    case kOk: return "no error";
    case kUnhandledCuda: return "unhandled HIP error (fake_rccl)";
    case kSystemError: return "system error: a peer did not arrive in time (fake_rccl)";
    case kInvalidArgument: return "invalid argument (fake_rccl)";
    case kInvalidUsage: return "invalid usage (fake_rccl)";
  }
  return "internal error (fake_rccl)";
}

int ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return kInvalidArgument;
  memset(id->internal, 0, sizeof id->internal);
  timespec t;
  clock_gettime(CLOCK_REALTIME, &t);
  snprintf(id->internal, sizeof id->internal, "/danhip_fake_rccl_%d_%lx%lx", (int)getpid(), (unsigned long)t.tv_sec, (unsigned long)t.tv_nsec);
  return kOk;
}

int ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/') return kInvalidArgument;
  Comm* c = new Comm;
  c->nranks = nranks;
  c->rank = rank;
  c->timeout_s = env_seconds("FAKE_RCCL_TIMEOUT_S", 120.0);
  c->slot_bytes = (size_t)env_seconds("FAKE_RCCL_SLOT_MB", 8.0) << 20;
  int dev = -1;
  if (hipGetDevice(&dev) == hipSuccess) c->device = dev;
  (void)hipGetLastError();
  strncpy(c->name, id.internal, sizeof c->name - 1);
  c->seg_bytes = kHeaderBytes + (size_t)nranks * c->slot_bytes;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { delete c; return kSystemError; }
  if (ftruncate(fd, (off_t)c->seg_bytes) != 0) { close(fd); delete c; return kSystemError; }
  c->seg = mmap(nullptr, c->seg_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->seg == MAP_FAILED) { c->seg = nullptr; delete c; return kSystemError; }
  c->hdr = (Header*)c->seg;
  c->hdr->attached.fetch_add(1);
  const bool hang = getenv("FAKE_RCCL_INIT_HANG") && atoi(getenv("FAKE_RCCL_INIT_HANG")) == 1;
  const bool ok = c->barrier(hang ? 0.0 : c->timeout_s);        // the bootstrap is a collective, as ncclCommInitRank is
  if (rank == 0 || !ok) shm_unlink(c->name);                     // every rank holds its mapping now: nothing stays in /dev/shm
  if (!ok) { release(c); return kSystemError; }
  *out = c;
  return kOk;
}

int ncclCommDestroy(ncclComm_t comm) {
  if (!comm) return kInvalidArgument;
  release((Comm*)comm);
  return kOk;
}

int ncclCommAbort(ncclComm_t comm) {
  if (!comm) return kInvalidArgument;
  Comm* c = (Comm*)comm;
  if (c->hdr) c->hdr->aborted.store(1);                          // peers waiting in a barrier return with an error
  // nothing is freed: a host callback of this communicator may be running on the runtime's thread right now, and the caller aborts
  // because it is about to end the process
  return kOk;
}

int ncclCommGetAsyncError(ncclComm_t comm, int* err) {
  if (!comm || !err) return kInvalidArgument;
  *err = ((Comm*)comm)->async_error.load();
  return kOk;
}

int ncclCommCount(ncclComm_t comm, int* n) {
  if (!comm || !n) return kInvalidArgument;
  *n = ((Comm*)comm)->nranks;
  return kOk;
}

int ncclCommUserRank(ncclComm_t comm, int* r) {
  if (!comm || !r) return kInvalidArgument;
  *r = ((Comm*)comm)->rank;
  return kOk;
}

int ncclCommCuDevice(ncclComm_t comm, int* d) {
  if (!comm || !d) return kInvalidArgument;
  *d = ((Comm*)comm)->device;
  return kOk;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, ncclComm_t comm, hipStream_t stream) {
  if (op != 0) return kInvalidArgument;                          // sum only
  return collective((Comm*)comm, kAllReduce, send, recv, count, dtype, stream);
}

int ncclReduceScatter(const void* send, void* recv, size_t recvcount, int dtype, int op, ncclComm_t comm, hipStream_t stream) {
  if (op != 0) return kInvalidArgument;
  return collective((Comm*)comm, kReduceScatter, send, recv, recvcount, dtype, stream);
}

int ncclAllGather(const void* send, void* recv, size_t sendcount, int dtype, ncclComm_t comm, hipStream_t stream) {
  return collective((Comm*)comm, kAllGather, send, recv, sendcount, dtype, stream);
}
}  // extern "C"
