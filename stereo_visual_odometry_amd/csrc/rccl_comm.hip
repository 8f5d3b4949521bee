// rccl_comm.hip -- a hipba_comm over a native RCCL communicator (ncclComm_t), host code only.
//
// The reference is a C++ program (src/solve/optimizer.cpp); a C++ caller of hipba_solve_device_sharded gets its
// all-reduce from here with no other runtime in the loop: every per-iteration sum of a sharded solve is one
// ncclAllReduce queued on the engine's own stream (in place, fp64), so nothing synchronises with the host.
// RCCL is loaded with dlopen on first use: a process that never shards a solve never loads it, and libhipba.so
// carries no link-time dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/hipba.h"

namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // the soname first: a process that already holds an RCCL (PyTorch's, say) shares that copy
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (!api.lib) return;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.lib, "ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.GetErrorString;
  });
  return api;
}

struct RcclComm {
  ncclComm_t comm = nullptr;
  std::atomic<long long> calls{0}, bytes{0};
  char last_error[256] = {0};
};

int rccl_allreduce(void* ctx, double* buf, size_t count, int op, void* stream) {
  auto* c = static_cast<RcclComm*>(ctx);
  const ncclResult_t r = rccl().AllReduce(buf, buf, count, ncclFloat64, op == HIPBA_REDUCE_MAX ? ncclMax : ncclSum, c->comm,
                                          static_cast<hipStream_t>(stream));
  if (r != ncclSuccess) {
    snprintf(c->last_error, sizeof(c->last_error), "ncclAllReduce: %s", rccl().GetErrorString(r));
    return 1;
  }
  c->calls.fetch_add(1, std::memory_order_relaxed);
  c->bytes.fetch_add((long long)(8 * count), std::memory_order_relaxed);
  return 0;
}

}  // namespace

extern "C" {

int hipba_rccl_get_unique_id(uint8_t id[HIPBA_RCCL_ID_BYTES]) {
  static_assert(HIPBA_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (!id) return HIPBA_ERR_INVALID_ARGUMENT;
  if (!rccl().ok) return HIPBA_ERR_COMM;
  ncclUniqueId u;
  if (rccl().GetUniqueId(&u) != ncclSuccess) return HIPBA_ERR_COMM;
  memcpy(id, u.internal, HIPBA_RCCL_ID_BYTES);
  return HIPBA_OK;
}

int hipba_rccl_comm_create(int device, const uint8_t id[HIPBA_RCCL_ID_BYTES], int32_t rank, int32_t world_size,
                           hipba_comm* out) {
  if (!id || !out || world_size < 1 || rank < 0 || rank >= world_size) return HIPBA_ERR_INVALID_ARGUMENT;
  memset(out, 0, sizeof(*out));
  if (!rccl().ok) return HIPBA_ERR_COMM;
  if (hipSetDevice(device) != hipSuccess) return HIPBA_ERR_HIP;
  RcclComm* c = new (std::nothrow) RcclComm();
  if (!c) return HIPBA_ERR_OUT_OF_MEMORY;
  ncclUniqueId u;
  memcpy(u.internal, id, HIPBA_RCCL_ID_BYTES);
  if (rccl().CommInitRank(&c->comm, world_size, u, rank) != ncclSuccess) {
    delete c;
    return HIPBA_ERR_COMM;
  }
  out->allreduce = &rccl_allreduce;
  out->ctx = c;
  out->rank = rank;
  out->world_size = world_size;
  return HIPBA_OK;
}

int hipba_rccl_comm_stats(const hipba_comm* comm, int64_t* calls, int64_t* bytes) {
  if (!comm || comm->allreduce != &rccl_allreduce || !comm->ctx) return HIPBA_ERR_INVALID_ARGUMENT;
  const auto* c = static_cast<const RcclComm*>(comm->ctx);
  if (calls) *calls = c->calls.load();
  if (bytes) *bytes = c->bytes.load();
  return HIPBA_OK;
}

int hipba_rccl_comm_destroy(hipba_comm* comm) {
  if (!comm || !comm->ctx) return HIPBA_OK;
  if (comm->allreduce != &rccl_allreduce) return HIPBA_ERR_INVALID_ARGUMENT;  // not one of ours
  auto* c = static_cast<RcclComm*>(comm->ctx);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  delete c;
  memset(comm, 0, sizeof(*comm));
  return HIPBA_OK;
}

}  // extern "C"
