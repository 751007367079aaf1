// comm.hip -- the gradient all-reduce of the data-parallel step behind the C ABI (include/w2v2_hip.h, "collective").
//
// ref: config/trainer/trainer.yaml:6-12 (PL `accelerator: ddp`): one process per GPU, the SUM all-reduce of the
// gradients is the only collective of a step.  The default binding of this repo issues it through torch.distributed
// ("nccl" == RCCL on ROCm, trainer.BucketAllReducer); these entry points let a caller WITHOUT torch.distributed run the
// same step: RCCL (librccl.so) is resolved at run time with dlopen -- libw2v2hip.so has no link-time dependency on it,
// and a process that already loaded an RCCL (torch's own copy) keeps using that one.
#include "common.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

namespace {
struct UniqueId { char internal[128]; };        // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                              // ncclComm_t
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*BroadcastFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllReduceFn all_reduce = nullptr;
  BroadcastFn broadcast = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GetErrorStringFn error_string = nullptr;
};
Rccl g_rccl;

int rccl_load() {
  if (g_rccl.handle != nullptr) return 0;
  void* h = nullptr;
  const char* names[] = {"librccl.so", "librccl.so.1"};
  for (const char* n : names)                                  // an RCCL this process already uses (e.g. torch's)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;
  if (h == nullptr) {
    const char* env = getenv("W2V2_RCCL_LIB");                 // explicit path
    if (env != nullptr) h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
  }
  if (h == nullptr) {
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
  }
  if (h == nullptr) {
    const char* rocm = getenv("ROCM_PATH");
    char path[512];
    snprintf(path, sizeof(path), "%s/lib/librccl.so", rocm != nullptr ? rocm : "/opt/rocm");
    h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  }
  if (h == nullptr) W2V2_FAIL("comm: cannot load librccl.so (%s); set W2V2_RCCL_LIB", dlerror());
  g_rccl.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
  g_rccl.all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
  g_rccl.broadcast = (BroadcastFn)dlsym(h, "ncclBroadcast");
  g_rccl.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
  g_rccl.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
  if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_reduce || !g_rccl.comm_destroy)
    W2V2_FAIL("comm: librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
  g_rccl.handle = h;
  return 0;
}
const char* rccl_err(int rc) { return g_rccl.error_string != nullptr ? g_rccl.error_string(rc) : "?"; }
}  // namespace

struct w2v2_comm {
  Comm comm;
  int rank, world, device;
};

extern "C" int w2v2_comm_unique_id(void* id_host_128) {
  W2V2_REQUIRE(id_host_128 != nullptr, "comm_unique_id: null buffer");
  if (rccl_load()) return -1;
  UniqueId id;
  const int rc = g_rccl.get_unique_id(&id);
  if (rc != 0) W2V2_FAIL("comm_unique_id: ncclGetUniqueId: %s", rccl_err(rc));
  memcpy(id_host_128, &id, sizeof(id));
  return 0;
}

extern "C" int w2v2_comm_init(w2v2_comm** out, const void* id_host_128, int rank, int world, int device) {
  W2V2_REQUIRE(out != nullptr && id_host_128 != nullptr && world >= 1 && rank >= 0 && rank < world && device >= 0,
               "comm_init: bad arguments (rank %d of %d, device %d)", rank, world, device);
  if (rccl_load()) return -1;
  // the communicator is bound to `device`; the caller's current device is restored
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (hipSetDevice(device) != hipSuccess) W2V2_FAIL("comm_init: hipSetDevice(%d) failed", device);
  w2v2_comm* w = (w2v2_comm*)malloc(sizeof(w2v2_comm));       // before the collective: nothing to unwind on failure
  if (w == nullptr) {
    if (prev >= 0) (void)hipSetDevice(prev);
    W2V2_FAIL("comm_init: out of memory");
  }
  UniqueId id;
  memcpy(&id, id_host_128, sizeof(id));
  Comm c = nullptr;
  const int rc = g_rccl.comm_init_rank(&c, world, id, rank);
  if (prev >= 0) (void)hipSetDevice(prev);
  if (rc != 0) {
    free(w);
    W2V2_FAIL("comm_init: ncclCommInitRank: %s", rccl_err(rc));
  }
  w->comm = c; w->rank = rank; w->world = world; w->device = device;
  *out = w;
  return 0;
}

extern "C" int w2v2_allreduce_async(w2v2_comm* comm, float* buf, int64_t n, void* stream) {
  W2V2_REQUIRE(comm != nullptr && buf != nullptr && n >= 0, "allreduce_async: bad arguments");
  if (n == 0) return 0;
  const int rc = g_rccl.all_reduce(buf, buf, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm->comm, as_stream(stream));
  if (rc != 0) W2V2_FAIL("allreduce_async: ncclAllReduce: %s", rccl_err(rc));
  return 0;
}

extern "C" int w2v2_broadcast_async(w2v2_comm* comm, void* buf, int64_t nbytes, int root, void* stream) {
  W2V2_REQUIRE(comm != nullptr && buf != nullptr && nbytes >= 0 && root >= 0 && root < comm->world,
               "broadcast_async: bad arguments (root %d)", root);
  if (nbytes == 0) return 0;
  W2V2_REQUIRE(g_rccl.broadcast != nullptr, "broadcast_async: librccl.so lacks ncclBroadcast");
  const int rc = g_rccl.broadcast(buf, buf, (size_t)nbytes, /*ncclUint8*/ 1, root, comm->comm, as_stream(stream));
  if (rc != 0) W2V2_FAIL("broadcast_async: ncclBroadcast: %s", rccl_err(rc));
  return 0;
}

extern "C" int w2v2_comm_destroy(w2v2_comm* comm) {
  if (comm == nullptr) return 0;
  const int rc = g_rccl.comm_destroy != nullptr ? g_rccl.comm_destroy(comm->comm) : 0;
  free(comm);
  if (rc != 0) W2V2_FAIL("comm_destroy: ncclCommDestroy: %s", rccl_err(rc));
  return 0;
}
