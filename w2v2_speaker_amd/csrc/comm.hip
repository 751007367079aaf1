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
  bool loopback;
};

// loop-back all-reduce: the SUM of `world` bit-identical contributions
__global__ __launch_bounds__(256) void loopback_scale_kernel(float* __restrict__ buf, int64_t n, float world) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) buf[i] *= world;
}

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
  w->comm = c; w->rank = rank; w->world = world; w->device = device; w->loopback = false;
  *out = w;
  return 0;
}

extern "C" int w2v2_comm_init_loopback(w2v2_comm** out, int world, int device) {
  W2V2_REQUIRE(out != nullptr && world >= 1 && device >= 0, "comm_init_loopback: bad arguments (world %d, device %d)", world,
               device);
  w2v2_comm* w = (w2v2_comm*)malloc(sizeof(w2v2_comm));
  if (w == nullptr) W2V2_FAIL("comm_init_loopback: out of memory");
  w->comm = nullptr; w->rank = 0; w->world = world; w->device = device; w->loopback = true;
  *out = w;
  return 0;
}

extern "C" int w2v2_allreduce_async(w2v2_comm* comm, float* buf, int64_t n, void* stream) {
  W2V2_REQUIRE(comm != nullptr && buf != nullptr && n >= 0, "allreduce_async: bad arguments");
  if (n == 0) return 0;
  if (comm->loopback) {
    if (comm->world > 1) {
      const int64_t blocks = (n + 255) / 256;
      hipLaunchKernelGGL(loopback_scale_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(stream),
                         buf, n, (float)comm->world);
      W2V2_CHECK_LAUNCH("allreduce_async (loop-back)");
    }
    return 0;
  }
  const int rc = g_rccl.all_reduce(buf, buf, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm->comm, as_stream(stream));
  if (rc != 0) W2V2_FAIL("allreduce_async: ncclAllReduce: %s", rccl_err(rc));
  return 0;
}

extern "C" int w2v2_broadcast_async(w2v2_comm* comm, void* buf, int64_t nbytes, int root, void* stream) {
  W2V2_REQUIRE(comm != nullptr && buf != nullptr && nbytes >= 0 && root >= 0 && root < comm->world,
               "broadcast_async: bad arguments (root %d)", root);
  if (nbytes == 0 || comm->loopback) return 0;
  W2V2_REQUIRE(g_rccl.broadcast != nullptr, "broadcast_async: librccl.so lacks ncclBroadcast");
  const int rc = g_rccl.broadcast(buf, buf, (size_t)nbytes, /*ncclUint8*/ 1, root, comm->comm, as_stream(stream));
  if (rc != 0) W2V2_FAIL("broadcast_async: ncclBroadcast: %s", rccl_err(rc));
  return 0;
}

extern "C" int w2v2_comm_destroy(w2v2_comm* comm) {
  if (comm == nullptr) return 0;
  const int rc = (!comm->loopback && g_rccl.comm_destroy != nullptr) ? g_rccl.comm_destroy(comm->comm) : 0;
  free(comm);
  if (rc != 0) W2V2_FAIL("comm_destroy: ncclCommDestroy: %s", rccl_err(rc));
  return 0;
}

// ------------------------------------------------------------------------------------------ overlap rehearsal (tools)
// A stand-in for the RCCL channels of a gradient all-reduce on a ONE-GPU box (tools/overlap_rehearsal.py): `channels`
// workgroups of 256 threads, each holding `lds_bytes` of LDS (an RCCL channel's footprint: it cannot share a CU with a
// 128-144 KiB GEMM workgroup), stream `bytes` of a bucket through themselves (read the slice, add, write it back: the HBM
// side of a ring step) -- PACED to `gbps` GB/s over all channels (s_memtime wall clock), so that the stand-in occupies its
// CUs for as long as a collective of that size would on xGMI.  What it measures is the compute-side cost of the
// collective's residency; it moves no data between GPUs.
__global__ __launch_bounds__(256) void traffic_probe_kernel(float* __restrict__ buf, int64_t n, int64_t chunk,
                                                            int64_t cycles_per_chunk) {
  extern __shared__ char probe_lds[];
  if (threadIdx.x == 0) probe_lds[0] = 0;                       // (keeps the dynamic LDS allocation)
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();          // constant 100 MHz counter
  int64_t done = 0;
  for (int64_t c0 = lo; c0 < hi; c0 += chunk, ++done) {
    const int64_t c1 = c0 + chunk < hi ? c0 + chunk : hi;
    for (int64_t i = c0 + threadIdx.x * 4; i + 3 < c1; i += 256 * 4) {
      float4 v = *reinterpret_cast<const float4*>(buf + i);
      v.x += 0.f; v.y += 0.f; v.z += 0.f; v.w += 0.f;
      *reinterpret_cast<float4*>(buf + i) = v;
    }
    if (cycles_per_chunk > 0) {
      const uint64_t due = t0 + (uint64_t)((done + 1) * cycles_per_chunk);
      while (__builtin_amdgcn_s_memrealtime() < due) __builtin_amdgcn_s_sleep(8);
    }
  }
}

extern "C" int w2v2_traffic_probe(float* buf, int64_t n, int channels, int lds_bytes, float gbps, void* stream) {
  W2V2_REQUIRE(buf != nullptr && n >= 0 && channels > 0 && channels <= 1024 && lds_bytes >= 0 && lds_bytes <= 160 * 1024,
               "traffic_probe: bad arguments");
  if (n == 0) return 0;
  W2V2_REQUIRE((reinterpret_cast<uintptr_t>(buf) & 15) == 0, "traffic_probe: 16-byte aligned buffer");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&traffic_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_set = true;
  }
  const int64_t chunk = 16384;                                  // floats per pacing step (64 KiB)
  // s_memrealtime counts at a fixed 100 MHz; one pacing step of every channel = channels * 64 KiB of the bucket
  int64_t cycles = 0;
  if (gbps > 0.f) cycles = (int64_t)(1e8 * (double)channels * (double)chunk * 4.0 / ((double)gbps * 1e9));    // gbps = bucket bytes per second
  hipLaunchKernelGGL(traffic_probe_kernel, dim3((unsigned)channels), dim3(256), (size_t)(lds_bytes > 16 ? lds_bytes : 16),
                     as_stream(stream), buf, n, chunk, cycles);
  W2V2_CHECK_LAUNCH("traffic_probe");
  return 0;
}
