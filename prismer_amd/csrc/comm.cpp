// libprismer_comm.so: gradient-bucket all-reduce over RCCL behind the C ABI of include/prismer_comm.h.
// One communicator per process (one process per GPU); the bf16 buckets of prismer_amd/dist.py travel through
// ph_allreduce_bucket on the Trainer's communication stream.  RCCL entry points are resolved with dlsym from the librccl
// already mapped into the process (PyTorch-ROCm bundles one; loading a second copy would split the device state), falling
// back to the system library.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "../../include/prismer_comm.h"

namespace {
thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

struct Api {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  const char* (*GetErrorString)(ncclResult_t);
  bool ok = false;
};

Api& api() {
  static Api a;
  static bool tried = false;
  if (tried) return a;
  tried = true;
  void* h = nullptr;
  const char* names[] = {"librccl.so", "librccl.so.1"};
  for (const char* n : names) {                       // a copy the process already holds (torch's) wins
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    const char* sys[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : sys) {
      h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
  }
  if (!h) return a;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
  a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
  a.ReduceScatter = (decltype(a.ReduceScatter))dlsym(h, "ncclReduceScatter");
  a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
  a.Broadcast = (decltype(a.Broadcast))dlsym(h, "ncclBroadcast");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  a.ok = a.GetUniqueId && a.CommInitRank && a.AllReduce && a.ReduceScatter && a.AllGather && a.Broadcast && a.CommDestroy && a.GetErrorString;
  return a;
}
}  // namespace

struct ph_comm {
  ncclComm_t comm;
  int rank, world;
};

static_assert(sizeof(ncclUniqueId) == PH_COMM_UNIQUE_ID_BYTES, "RCCL unique id size changed");

#define NEED_RCCL()                                                                                             \
  Api& A = api();                                                                                               \
  if (!A.ok) return fail(PH_COMM_ERR_NO_RCCL, "librccl not found (neither mapped in the process nor on the loader path)")
#define RCCL_CALL(expr, what)                                                                                   \
  do {                                                                                                          \
    ncclResult_t r__ = (expr);                                                                                  \
    if (r__ != ncclSuccess) return fail(PH_COMM_ERR_RCCL, "%s: %s", what, A.GetErrorString(r__));               \
  } while (0)

extern "C" const char* ph_comm_last_error(void) { return g_err.c_str(); }

extern "C" int ph_comm_unique_id(void* out_id_128) {
  if (!out_id_128) return fail(PH_COMM_ERR_BAD_ARG, "ph_comm_unique_id: null output");
  NEED_RCCL();
  ncclUniqueId id;
  RCCL_CALL(A.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(out_id_128, &id, sizeof(id));
  return PH_COMM_OK;
}

extern "C" int ph_comm_init(int rank, int world, const void* unique_id_128, ph_comm** out) {
  if (!out || !unique_id_128 || world < 1 || rank < 0 || rank >= world)
    return fail(PH_COMM_ERR_BAD_ARG, "ph_comm_init: bad arguments (rank %d of %d)", rank, world);
  NEED_RCCL();
  ncclUniqueId id;
  memcpy(&id, unique_id_128, sizeof(id));
  ncclComm_t c;
  RCCL_CALL(A.CommInitRank(&c, world, id, rank), "ncclCommInitRank");
  *out = new ph_comm{c, rank, world};
  return PH_COMM_OK;
}

extern "C" int ph_allreduce_bucket(ph_comm* comm, void* buf, int64_t count, int dtype, hipStream_t stream) {
  if (!comm || !buf || count < 0) return fail(PH_COMM_ERR_BAD_ARG, "ph_allreduce_bucket: bad arguments");
  if (dtype != PH_COMM_F32 && dtype != PH_COMM_BF16) return fail(PH_COMM_ERR_BAD_ARG, "ph_allreduce_bucket: dtype %d", dtype);
  if (count == 0) return PH_COMM_OK;
  NEED_RCCL();
  RCCL_CALL(A.AllReduce(buf, buf, (size_t)count, dtype == PH_COMM_BF16 ? ncclBfloat16 : ncclFloat32, ncclSum, comm->comm, stream),
            "ncclAllReduce");
  return PH_COMM_OK;
}

static int dtype_of(int dtype, ncclDataType_t* out) {
  if (dtype == PH_COMM_F32) { *out = ncclFloat32; return 0; }
  if (dtype == PH_COMM_BF16) { *out = ncclBfloat16; return 0; }
  return -1;
}

extern "C" int ph_reduce_scatter(ph_comm* comm, const void* send, void* recv, int64_t recv_count, int dtype, hipStream_t stream) {
  ncclDataType_t dt;
  if (!comm || !send || !recv || recv_count < 0 || dtype_of(dtype, &dt)) return fail(PH_COMM_ERR_BAD_ARG, "ph_reduce_scatter: bad arguments");
  if (recv_count == 0) return PH_COMM_OK;
  NEED_RCCL();
  RCCL_CALL(A.ReduceScatter(send, recv, (size_t)recv_count, dt, ncclSum, comm->comm, stream), "ncclReduceScatter");
  return PH_COMM_OK;
}

extern "C" int ph_all_gather(ph_comm* comm, const void* send, void* recv, int64_t send_count, int dtype, hipStream_t stream) {
  ncclDataType_t dt;
  if (!comm || !send || !recv || send_count < 0 || dtype_of(dtype, &dt)) return fail(PH_COMM_ERR_BAD_ARG, "ph_all_gather: bad arguments");
  if (send_count == 0) return PH_COMM_OK;
  NEED_RCCL();
  RCCL_CALL(A.AllGather(send, recv, (size_t)send_count, dt, comm->comm, stream), "ncclAllGather");
  return PH_COMM_OK;
}

extern "C" int ph_broadcast(ph_comm* comm, void* buf, int64_t count, int dtype, int root, hipStream_t stream) {
  ncclDataType_t dt;
  if (!comm || !buf || count < 0 || dtype_of(dtype, &dt) || root < 0 || root >= comm->world) return fail(PH_COMM_ERR_BAD_ARG, "ph_broadcast: bad arguments");
  if (count == 0) return PH_COMM_OK;
  NEED_RCCL();
  RCCL_CALL(A.Broadcast(buf, buf, (size_t)count, dt, root, comm->comm, stream), "ncclBroadcast");
  return PH_COMM_OK;
}

extern "C" int ph_comm_world(const ph_comm* comm) { return comm ? comm->world : 0; }

extern "C" int ph_comm_destroy(ph_comm* comm) {
  if (!comm) return PH_COMM_OK;
  NEED_RCCL();
  ncclResult_t r = A.CommDestroy(comm->comm);
  delete comm;
  if (r != ncclSuccess) return fail(PH_COMM_ERR_RCCL, "ncclCommDestroy: %s", A.GetErrorString(r));
  return PH_COMM_OK;
}
