// egt_dp_* — the per-step gradient all-reduce of batch data parallelism, straight on RCCL.
//
// What tf.distribute.MirroredStrategy does for the reference (lib/training/training_base.py:230-247): one
// synchronous all-reduce of the gradients per step.  One process per GPU, ONE communicator per process, the
// collective is enqueued on the caller's stream (the stream the backward ran on: no host sync, no event hop).
// RCCL is resolved at run time (dlopen of the copy already mapped into the process — PyTorch ships one — else
// librccl.so.1 from the loader path), so libegt_amd.so carries no link-time dependency on it and a second RCCL
// copy is never pulled into a process that already has one.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include <rccl/rccl.h>   // types and enum values only: every entry point is resolved with dlsym (no link-time dependency)

#include "egt_common.h"

namespace {
// the slice of the RCCL ABI this file uses, taken from the installed header and pinned at build time: the C-ABI's
// EGT_DP_ID_BYTES (include/egt_amd.h) is what callers allocate for the id, and the enum values travel as plain ints
// through the dlsym'd function pointers below
static_assert(sizeof(ncclUniqueId) == EGT_DP_ID_BYTES, "include/egt_amd.h: EGT_DP_ID_BYTES != sizeof(ncclUniqueId) of <rccl/rccl.h>");
static_assert(NCCL_UNIQUE_ID_BYTES == EGT_DP_ID_BYTES, "NCCL_UNIQUE_ID_BYTES changed");
enum { kNcclSuccess = ncclSuccess, kNcclFloat32 = ncclFloat32, kNcclSum = ncclSum, kNcclAvg = ncclAvg };
static_assert(kNcclSuccess == 0 && kNcclFloat32 == 7 && kNcclSum == 0 && kNcclAvg == 4, "RCCL enum values moved (documented in INTEGRATION.md)");
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int),
              "the function-pointer signatures below pass RCCL's enums as int");

struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

std::mutex g_mu;
Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_world = 0, g_rank = -1;

bool load_rccl() {
  if (g_rccl.so) return true;
  const char* names[] = {"librccl.so", "librccl.so.1"};
  void* so = nullptr;
  for (const char* n : names) if ((so = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;   // a copy the process already has
  if (!so) for (const char* n : names) if ((so = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!so) so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!so) { egt_set_error("RCCL not found: %s", dlerror()); return false; }
  Rccl r;
  r.so = so;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(so, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(so, "ncclCommInitRank");
  r.AllReduce = (decltype(r.AllReduce))dlsym(so, "ncclAllReduce");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(so, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(so, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) {
    egt_set_error("RCCL library lacks a required symbol");
    return false;
  }
  g_rccl = r;
  return true;
}

int fail_rccl(const char* what, int rc) {
  egt_set_error("%s failed: %s (%d)", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?", rc);
  return EGT_E_RCCL;
}
}  // namespace

extern "C" int egt_dp_unique_id(void* id_out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!id_out) EGT_FAIL(EGT_E_NULL, "id_out is NULL");
  if (!load_rccl()) return EGT_E_RCCL;
  ncclUniqueId id;
  const int rc = g_rccl.GetUniqueId(&id);
  if (rc != kNcclSuccess) return fail_rccl("ncclGetUniqueId", rc);
  memcpy(id_out, &id, EGT_DP_ID_BYTES);
  return EGT_OK;
}

extern "C" int egt_dp_init(const void* id_in, int32_t world, int32_t rank) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!id_in) EGT_FAIL(EGT_E_NULL, "id is NULL");
  if (world < 1 || rank < 0 || rank >= world) EGT_FAIL(EGT_E_SHAPE, "bad world/rank %d/%d", world, rank);
  if (g_comm) EGT_FAIL(EGT_E_FLAGS, "egt_dp_init: a communicator already exists (one per process); call egt_dp_finalize first");
  if (!load_rccl()) return EGT_E_RCCL;
  ncclUniqueId id;
  memcpy(&id, id_in, EGT_DP_ID_BYTES);
  const int rc = g_rccl.CommInitRank(&g_comm, world, id, rank);   // binds the CURRENT HIP device of the calling thread
  if (rc != kNcclSuccess) { g_comm = nullptr; return fail_rccl("ncclCommInitRank", rc); }
  g_world = world;
  g_rank = rank;
  return EGT_OK;
}

extern "C" int egt_dp_allreduce(void* buf, size_t count, int32_t average, void* stream) {
  if (!g_comm) EGT_FAIL(EGT_E_FLAGS, "egt_dp_allreduce before egt_dp_init");
  if (!buf && count) EGT_FAIL(EGT_E_NULL, "buf is NULL");
  if (!count) return EGT_OK;
  const int rc = g_rccl.AllReduce(buf, buf, count, kNcclFloat32, average ? kNcclAvg : kNcclSum, g_comm, (hipStream_t)stream);
  if (rc != kNcclSuccess) return fail_rccl("ncclAllReduce", rc);
  return EGT_OK;
}

extern "C" int egt_dp_world(void) { return g_world; }
extern "C" int egt_dp_rank(void) { return g_rank; }

extern "C" int egt_dp_finalize(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_comm) return EGT_OK;
  const int rc = g_rccl.CommDestroy(g_comm);
  g_comm = nullptr;
  g_world = 0;
  g_rank = -1;
  if (rc != kNcclSuccess) return fail_rccl("ncclCommDestroy", rc);
  return EGT_OK;
}
