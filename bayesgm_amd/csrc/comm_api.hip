// comm_api.hip -- the RCCL communicator behind the C ABI: what the data-parallel minibatch loop inside the library
// (bgm_causal_fit_epoch_dp, fit_api.hip; bgm_bnn_fit_epoch_dp, bnn_api.hip) enqueues its gradient all-reduce on.
// replaces: nothing in the reference (it has no collective); the loop that is sharded is causalbgm/base.py:488-514 and the
// exchange is north_star's "RCCL all-reduce of generator / discriminator gradients over xGMI" (one fused g|f|h buffer per step).
//
// RCCL is resolved at RUN time: the process that calls this already holds a librccl (PyTorch-ROCm links one: SONAME librccl.so.1), and two
// copies of a collective library in one process is what must not happen -- so the library has no link-time dependency, dlopen() asks for the
// copy already mapped first (RTLD_NOLOAD) and only then for the system one.  Only the types of <rccl/rccl.h> are used at compile time.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include <rccl/rccl.h>

#include "bgm_host.h"
#include "comm_host.h"

namespace {
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string where, error;
};

RcclApi &rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *env = std::getenv("BGM_RCCL_LIB");
    const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int pass = 0; pass < 2 && !api.lib; ++pass)          // pass 0: a copy the process already holds; pass 1: load one
      for (const char *n : names) {
        if (!n || !*n) continue;
        api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (api.lib) { api.where = std::string(n) + (pass == 0 ? " (already mapped)" : " (loaded)"); break; }
      }
    if (!api.lib) { api.error = std::string("RCCL not found (librccl.so.1; set BGM_RCCL_LIB): ") + (dlerror() ? dlerror() : ""); return; }
    auto sym = [&](const char *name) -> void * {
      void *p = dlsym(api.lib, name);
      if (!p && api.error.empty()) api.error = std::string("RCCL symbol missing: ") + name;
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
    api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return api;
}

int rccl_ready(const char *who) {
  RcclApi &a = rccl();
  if (!a.error.empty()) { bgm_set_error(std::string(who) + ": " + a.error); return BGM_E_UNSUPPORTED; }
  return BGM_OK;
}

int rccl_fail(const char *who, ncclResult_t r) {
  RcclApi &a = rccl();
  bgm_set_error(std::string(who) + ": " + (a.GetErrorString ? a.GetErrorString(r) : "RCCL error") + " (" + std::to_string((int)r) + ")");
  return BGM_E_HIP;
}
}  // namespace

// sum-all-reduce of `count` floats in place on `stream` (ordered with the kernels around it like any launch); comm = ncclComm_t
int bgm_comm_enqueue_all_reduce(void *comm, float *buf, long long count, hipStream_t stream) {
  int rc = rccl_ready("all-reduce");
  if (rc) return rc;
  if (!comm || !buf || count < 0) { bgm_set_error("all-reduce: NULL communicator / buffer"); return BGM_E_INVALID; }
  if (count == 0) return BGM_OK;
  const ncclResult_t r = rccl().AllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, (ncclComm_t)comm, stream);
  return r == ncclSuccess ? BGM_OK : rccl_fail("ncclAllReduce", r);
}

int bgm_comm_world(void *comm, int *world, int *rank) {
  int rc = rccl_ready("communicator");
  if (rc) return rc;
  if (!comm) { bgm_set_error("communicator is NULL"); return BGM_E_INVALID; }
  ncclResult_t r = rccl().CommCount((ncclComm_t)comm, world);
  if (r == ncclSuccess && rank) r = rccl().CommUserRank((ncclComm_t)comm, rank);
  return r == ncclSuccess ? BGM_OK : rccl_fail("ncclCommCount", r);
}

extern "C" int bgm_comm_unique_id(void *id128) {
  static_assert(sizeof(ncclUniqueId) == BGM_COMM_ID_BYTES, "bgm_hip.h: BGM_COMM_ID_BYTES");
  int rc = rccl_ready("bgm_comm_unique_id");
  if (rc) return rc;
  if (!id128) { bgm_set_error("bgm_comm_unique_id: NULL output"); return BGM_E_INVALID; }
  ncclUniqueId id;
  const ncclResult_t r = rccl().GetUniqueId(&id);
  if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
  std::memcpy(id128, &id, sizeof(id));
  return BGM_OK;
}

extern "C" int bgm_comm_create(int32_t device, const void *id128, int32_t world, int32_t rank, void **comm_out) {
  int rc = rccl_ready("bgm_comm_create");
  if (rc) return rc;
  if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world) { bgm_set_error("bgm_comm_create: bad argument"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(device));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  const ncclResult_t r = rccl().CommInitRank(&c, world, id, rank);
  if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
  *comm_out = (void *)c;
  return BGM_OK;
}

extern "C" int bgm_comm_destroy(void *comm) {
  if (!comm) return BGM_OK;
  int rc = rccl_ready("bgm_comm_destroy");
  if (rc) return rc;
  const ncclResult_t r = rccl().CommDestroy((ncclComm_t)comm);
  return r == ncclSuccess ? BGM_OK : rccl_fail("ncclCommDestroy", r);
}

extern "C" int bgm_comm_info(void *comm, int32_t *world, int32_t *rank, char *library, int32_t library_cap) {
  int w = 0, r = 0;
  int rc = bgm_comm_world(comm, &w, &r);
  if (rc) return rc;
  if (world) *world = w;
  if (rank) *rank = r;
  if (library && library_cap > 0) {
    std::strncpy(library, rccl().where.c_str(), (size_t)library_cap - 1);
    library[library_cap - 1] = 0;
  }
  return BGM_OK;
}

extern "C" int bgm_comm_all_reduce_f32(void *comm, float *buf_dev, int64_t count, void *stream) {
  return bgm_comm_enqueue_all_reduce(comm, buf_dev, (long long)count, (hipStream_t)stream);
}
