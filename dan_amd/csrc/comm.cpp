// Data-parallel gradient exchange on RCCL, called directly (include/danhip.h: danhip_comm_*).
//
// What it stands in for: tf_replicate_model_fn.py:633-645 (_compute_sum_on_device = add_n of the towers' gradients on one device) —
// on one-process-per-GPU ranks that sum is an all-reduce over xGMI.  The collectives are enqueued on the CALLER's stream, so the
// trainer orders them against its backward kernels with events and may record them into a hipGraph; nothing here owns a thread.
//
// librccl is bound at run time (dlopen + dlsym): processes that never exchange gradients do not need it, and inside a PyTorch process
// the SAME copy PyTorch loaded is reused (two RCCL copies in one process would each claim the device's IPC resources).
// Only the stable C entry points of rccl.h are used; their types are restated here (ncclUniqueId = 128 opaque bytes rccl.h:40-43,
// ncclDataType_t rccl.h:461-470, ncclRedOp_t rccl.h:448) so the build does not depend on which RCCL header version is installed.
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>

#include "common.h"

namespace {
typedef struct { char internal[DANHIP_COMM_ID_BYTES]; } rccl_id_t;
typedef void* rccl_comm_t;
enum { kNcclSum = 0, kNcclHalf = 6, kNcclFloat = 7, kNcclBfloat16 = 9 };

struct Rccl {
  void* handle = nullptr;
  int (*GetVersion)(int*) = nullptr;
  int (*GetUniqueId)(rccl_id_t*) = nullptr;
  int (*CommInitRank)(rccl_comm_t*, int, rccl_id_t, int) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*CommCount)(rccl_comm_t, int*) = nullptr;
  int (*CommUserRank)(rccl_comm_t, int*) = nullptr;
  int (*CommCuDevice)(rccl_comm_t, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
  // optional (rccl.h ncclCommGetAsyncError / ncclCommAbort): the failure-detection pair — a library without them reports "healthy"
  int (*CommGetAsyncError)(rccl_comm_t, int*) = nullptr;
  int (*CommAbort)(rccl_comm_t) = nullptr;
};
Rccl g_rccl;                       // written once under g_mu, published by g_bound (release) — read-only afterwards
std::mutex g_mu;
std::atomic<bool> g_bound{false};

template <class F>
bool sym(void* h, const char* name, F& out) {
  out = reinterpret_cast<F>(dlsym(h, name));
  return out != nullptr;
}

int bind(const char* path) {
  if (g_bound.load(std::memory_order_acquire)) return DANHIP_OK;
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_bound.load(std::memory_order_relaxed)) return DANHIP_OK;
  void* h = nullptr;
  if (path && *path) {
    h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    DH_REQUIRE(h, DANHIP_ECOMM, "danhip_comm_load: dlopen(%s) failed: %s", path, dlerror());
  } else {
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)                        // a copy already mapped into the process wins
      if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    const char* env = getenv("DANHIP_RCCL_PATH");
    if (!h && env && *env) h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    for (const char* n : names)
      if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    DH_REQUIRE(h, DANHIP_ECOMM, "danhip_comm: librccl.so not found (call danhip_comm_load(path) or set DANHIP_RCCL_PATH): %s", dlerror());
  }
  Rccl r;
  r.handle = h;
  const bool ok = sym(h, "ncclGetVersion", r.GetVersion) && sym(h, "ncclGetUniqueId", r.GetUniqueId) &&
                  sym(h, "ncclCommInitRank", r.CommInitRank) && sym(h, "ncclCommDestroy", r.CommDestroy) &&
                  sym(h, "ncclCommCount", r.CommCount) && sym(h, "ncclCommUserRank", r.CommUserRank) &&
                  sym(h, "ncclCommCuDevice", r.CommCuDevice) && sym(h, "ncclGetErrorString", r.GetErrorString) &&
                  sym(h, "ncclAllReduce", r.AllReduce) && sym(h, "ncclReduceScatter", r.ReduceScatter) && sym(h, "ncclAllGather", r.AllGather);
  DH_REQUIRE(ok, DANHIP_ECOMM, "danhip_comm: %s lacks an RCCL entry point: %s", path ? path : "librccl.so", dlerror());
  (void)sym(h, "ncclCommGetAsyncError", r.CommGetAsyncError);
  (void)sym(h, "ncclCommAbort", r.CommAbort);
  g_rccl = r;
  g_bound.store(true, std::memory_order_release);
  return DANHIP_OK;
}

#define DH_RCCL(call, what)                                                                           \
  do {                                                                                                \
    const int rc__ = (call);                                                                          \
    if (rc__ != 0) {                                                                                  \
      danhip_set_error("%s: RCCL error %d (%s)", what, rc__, g_rccl.GetErrorString(rc__));            \
      return DANHIP_ECOMM;                                                                            \
    }                                                                                                 \
  } while (0)

int nccl_dtype(int dtype, int* out, size_t* elem) {
  switch (dtype) {
    case DANHIP_F32: *out = kNcclFloat; *elem = 4; return DANHIP_OK;
    case DANHIP_BF16: *out = kNcclBfloat16; *elem = 2; return DANHIP_OK;
    case DANHIP_F16: *out = kNcclHalf; *elem = 2; return DANHIP_OK;
  }
  danhip_set_error("danhip_comm: dtype %d is not DANHIP_F32 / DANHIP_BF16 / DANHIP_F16", dtype);
  return DANHIP_EINVAL;
}
}  // namespace

extern "C" int danhip_comm_load(const char* librccl_path) { return bind(librccl_path); }

extern "C" int danhip_comm_rccl_version(int* version) {
  DH_REQUIRE(version, DANHIP_EINVAL, "danhip_comm_rccl_version: NULL");
  if (int rc = bind(nullptr)) return rc;
  DH_RCCL(g_rccl.GetVersion(version), "ncclGetVersion");
  return DANHIP_OK;
}

extern "C" int danhip_comm_unique_id(void* id128) {
  DH_REQUIRE(id128, DANHIP_EINVAL, "danhip_comm_unique_id: NULL");
  if (int rc = bind(nullptr)) return rc;
  rccl_id_t id;
  DH_RCCL(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id128, id.internal, DANHIP_COMM_ID_BYTES);
  return DANHIP_OK;
}

extern "C" int danhip_comm_create(const void* id128, int32_t nranks, int32_t rank, int32_t device, void** comm_out) {
  DH_REQUIRE(id128 && comm_out, DANHIP_EINVAL, "danhip_comm_create: NULL argument");
  DH_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, DANHIP_EINVAL, "danhip_comm_create: rank %d of %d", rank, nranks);
  if (int rc = bind(nullptr)) return rc;
  if (device >= 0) {                                   // RCCL binds the communicator to the calling thread's current device
    const hipError_t e = hipSetDevice(device);
    DH_REQUIRE(e == hipSuccess, DANHIP_EINVAL, "danhip_comm_create: hipSetDevice(%d): %s", device, hipGetErrorString(e));
  }
  rccl_id_t id;
  memcpy(id.internal, id128, DANHIP_COMM_ID_BYTES);
  rccl_comm_t c = nullptr;
  DH_RCCL(g_rccl.CommInitRank(&c, nranks, id, rank), "ncclCommInitRank");
  *comm_out = c;
  return DANHIP_OK;
}

extern "C" int danhip_comm_destroy(void* comm) {
  if (!comm) return DANHIP_OK;
  DH_REQUIRE(g_bound.load(std::memory_order_acquire), DANHIP_ECOMM, "danhip_comm_destroy: RCCL was never bound");
  DH_RCCL(g_rccl.CommDestroy(comm), "ncclCommDestroy");
  return DANHIP_OK;
}

extern "C" int danhip_comm_abort(void* comm) {
  if (!comm) return DANHIP_OK;
  DH_REQUIRE(g_bound.load(std::memory_order_acquire), DANHIP_ECOMM, "danhip_comm_abort: RCCL was never bound");
  if (g_rccl.CommAbort) {
    DH_RCCL(g_rccl.CommAbort(comm), "ncclCommAbort");
  } else {
    DH_RCCL(g_rccl.CommDestroy(comm), "ncclCommDestroy");
  }
  return DANHIP_OK;
}

extern "C" int danhip_comm_async_error(void* comm, int32_t* err) {
  DH_REQUIRE(comm && err && g_bound.load(std::memory_order_acquire), DANHIP_EINVAL, "danhip_comm_async_error: no communicator");
  *err = 0;
  if (!g_rccl.CommGetAsyncError) return DANHIP_OK;
  int e = 0;
  DH_RCCL(g_rccl.CommGetAsyncError(comm, &e), "ncclCommGetAsyncError");
  *err = e;
  if (e != 0) danhip_set_error("danhip_comm: asynchronous RCCL error %d (%s)", e, g_rccl.GetErrorString(e));
  return DANHIP_OK;
}

extern "C" int danhip_comm_info(void* comm, int32_t* nranks, int32_t* rank, int32_t* device) {
  DH_REQUIRE(comm && g_bound.load(std::memory_order_acquire), DANHIP_EINVAL, "danhip_comm_info: no communicator");
  int v = 0;
  if (nranks) { DH_RCCL(g_rccl.CommCount(comm, &v), "ncclCommCount"); *nranks = v; }
  if (rank) { DH_RCCL(g_rccl.CommUserRank(comm, &v), "ncclCommUserRank"); *rank = v; }
  if (device) { DH_RCCL(g_rccl.CommCuDevice(comm, &v), "ncclCommCuDevice"); *device = v; }
  return DANHIP_OK;
}

extern "C" int danhip_comm_allreduce_sum(void* comm, void* buf, int64_t count, int dtype, void* stream) {
  DH_REQUIRE(comm && g_bound.load(std::memory_order_acquire), DANHIP_EINVAL, "danhip_comm_allreduce_sum: no communicator");
  DH_REQUIRE(count >= 0 && (buf || count == 0), DANHIP_EINVAL, "danhip_comm_allreduce_sum: bad buffer");
  int dt; size_t es;
  if (int rc = nccl_dtype(dtype, &dt, &es)) return rc;
  if (count == 0) return DANHIP_OK;
  DH_RCCL(g_rccl.AllReduce(buf, buf, (size_t)count, dt, kNcclSum, comm, (hipStream_t)stream), "ncclAllReduce");
  return DANHIP_OK;
}

extern "C" int danhip_comm_reduce_scatter_sum(void* comm, const void* send, void* recv, int64_t recvcount, int dtype, void* stream) {
  DH_REQUIRE(comm && g_bound.load(std::memory_order_acquire), DANHIP_EINVAL, "danhip_comm_reduce_scatter_sum: no communicator");
  DH_REQUIRE(recvcount >= 0 && ((send && recv) || recvcount == 0), DANHIP_EINVAL, "danhip_comm_reduce_scatter_sum: bad buffer");
  int dt; size_t es;
  if (int rc = nccl_dtype(dtype, &dt, &es)) return rc;
  if (recvcount == 0) return DANHIP_OK;
  DH_RCCL(g_rccl.ReduceScatter(send, recv, (size_t)recvcount, dt, kNcclSum, comm, (hipStream_t)stream), "ncclReduceScatter");
  return DANHIP_OK;
}

extern "C" int danhip_comm_allgather(void* comm, const void* send, void* recv, int64_t sendcount, int dtype, void* stream) {
  DH_REQUIRE(comm && g_bound.load(std::memory_order_acquire), DANHIP_EINVAL, "danhip_comm_allgather: no communicator");
  DH_REQUIRE(sendcount >= 0 && ((send && recv) || sendcount == 0), DANHIP_EINVAL, "danhip_comm_allgather: bad buffer");
  int dt; size_t es;
  if (int rc = nccl_dtype(dtype, &dt, &es)) return rc;
  if (sendcount == 0) return DANHIP_OK;
  DH_RCCL(g_rccl.AllGather(send, recv, (size_t)sendcount, dt, comm, (hipStream_t)stream), "ncclAllGather");
  return DANHIP_OK;
}
