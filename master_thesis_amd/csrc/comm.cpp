// Data-parallel gradient exchange through the C ABI: an RCCL communicator owned
// by libmt355 (one rank per GPU, xGMI underneath) and an in-place fp32 SUM
// all-reduce issued on the caller's stream.  The reference has no collective of
// its own (Lightning's DDP would insert one after backward);
// this is the MI355X-native equivalent.
//
// RCCL is bound at run time (dlopen): the library loads, and every other entry
// point works, on a box without RCCL or without a second GPU.  The host side
// passes the path of the RCCL build its process already uses (torch's), so that
// one process never holds two copies.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "mt355.h"

void mt_set_error(const char* fmt, ...);

namespace {

// the slice of rccl.h this file needs (types are ABI-stable: NCCL 2.x)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclFloat32 = 7 };
enum { ncclSum = 0 };

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*CommCount)(const ncclComm_t, int*) = nullptr;
  int (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

template <class F>
bool bind(F& fn, const char* name) {
  fn = reinterpret_cast<F>(dlsym(g_rccl.handle, name));
  return fn != nullptr;
}

int load_rccl(const char* path) {
  if (g_rccl.handle) return MT_OK;
  const char* tries[] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* t : tries) {
    if (!t || !t[0]) continue;
    g_rccl.handle = dlopen(t, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) {
    mt_set_error("mt_comm: cannot load RCCL (%s)", dlerror());
    return MT_ERR_LAUNCH;
  }
  const bool ok = bind(g_rccl.GetUniqueId, "ncclGetUniqueId") && bind(g_rccl.CommInitRank, "ncclCommInitRank") &&
                  bind(g_rccl.CommDestroy, "ncclCommDestroy") && bind(g_rccl.CommCount, "ncclCommCount") &&
                  bind(g_rccl.CommUserRank, "ncclCommUserRank") && bind(g_rccl.AllReduce, "ncclAllReduce") &&
                  bind(g_rccl.GetErrorString, "ncclGetErrorString");
  if (!ok) {
    mt_set_error("mt_comm: RCCL symbols missing in the loaded library");
    dlclose(g_rccl.handle);
    g_rccl = Rccl();
    return MT_ERR_LAUNCH;
  }
  return MT_OK;
}

int nccl_fail(const char* what, int rc) {
  mt_set_error("%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
  return MT_ERR_LAUNCH;
}

}  // namespace

extern "C" int mt_comm_unique_id(const char* rccl_path, void* id128) {
  if (!id128) { mt_set_error("mt_comm_unique_id: null output"); return MT_ERR_ARG; }
  int rc = load_rccl(rccl_path);
  if (rc) return rc;
  ncclUniqueId id;
  rc = g_rccl.GetUniqueId(&id);
  if (rc != ncclSuccess) return nccl_fail("ncclGetUniqueId", rc);
  memcpy(id128, id.internal, sizeof(id.internal));
  return MT_OK;
}

extern "C" int mt_comm_init(const char* rccl_path, const void* id128, int32_t rank, int32_t world, void** comm) {
  if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) {
    mt_set_error("mt_comm_init: bad arguments (rank %d of %d)", rank, world);
    return MT_ERR_ARG;
  }
  int rc = load_rccl(rccl_path);
  if (rc) return rc;
  ncclUniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  ncclComm_t c = nullptr;
  rc = g_rccl.CommInitRank(&c, world, id, rank);      // on the calling thread's current HIP device
  if (rc != ncclSuccess) return nccl_fail("ncclCommInitRank", rc);
  *comm = c;
  return MT_OK;
}

extern "C" int mt_comm_info(void* comm, int32_t* rank, int32_t* world) {
  if (!comm || !g_rccl.handle) { mt_set_error("mt_comm_info: no communicator"); return MT_ERR_ARG; }
  int r = 0, w = 0;
  int rc = g_rccl.CommUserRank(static_cast<ncclComm_t>(comm), &r);
  if (rc == ncclSuccess) rc = g_rccl.CommCount(static_cast<ncclComm_t>(comm), &w);
  if (rc != ncclSuccess) return nccl_fail("ncclCommCount", rc);
  if (rank) *rank = r;
  if (world) *world = w;
  return MT_OK;
}

extern "C" int mt_comm_destroy(void* comm) {
  if (!comm) return MT_OK;
  if (!g_rccl.handle) { mt_set_error("mt_comm_destroy: RCCL not loaded"); return MT_ERR_ARG; }
  const int rc = g_rccl.CommDestroy(static_cast<ncclComm_t>(comm));
  return rc == ncclSuccess ? MT_OK : nccl_fail("ncclCommDestroy", rc);
}

extern "C" int mt_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream) {
  if (!comm || !buf || n <= 0 || !g_rccl.handle) {
    mt_set_error("mt_allreduce_sum_f32: bad arguments");
    return MT_ERR_ARG;
  }
  const int rc = g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, static_cast<ncclComm_t>(comm),
                                  static_cast<hipStream_t>(stream));
  return rc == ncclSuccess ? MT_OK : nccl_fail("ncclAllReduce", rc);
}
