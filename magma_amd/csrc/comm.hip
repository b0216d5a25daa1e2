// comm.hip -- the data-parallel exchange step of the training path as C-ABI entry points over RCCL (xGMI inside a node):
// what a reference maintainer binds instead of DeepSpeed's gradient reduction (reference magma/train_loop.py:18-19,
// train.py:103-111; SURVEY 8b mg_comm_{init,allreduce,destroy}).
//
// RCCL is reached through dlopen / dlsym, not at link time: libmagma_hip.so stays loadable on a box without RCCL (the
// inference path never needs it), and inside a PyTorch process the library PyTorch has already loaded (torch/lib/librccl.so)
// is re-used instead of a second copy with its own communicator state.  The prototypes below restate the public RCCL API
// (rccl.h: NCCL 2.x compatible), nothing else of it is used.
//
// One communicator per process = per GPU; the caller distributes the 128-byte unique id of rank 0 (any side channel: here
// torch.distributed's store, in the reference it would be its launcher's), calls mg_comm_init on every rank, enqueues
// all-reduces on a stream of its choice (the engine uses a side stream so that they run under the backward pass) and orders
// them against its compute with events.  Enqueue-only, like every other entry point.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace {
typedef struct { char internal[128]; } rcclUniqueId;
typedef void* rcclComm_t;
typedef int (*pfn_GetUniqueId)(rcclUniqueId*);
typedef int (*pfn_CommInitRank)(rcclComm_t*, int, rcclUniqueId, int);
typedef int (*pfn_AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t);
typedef int (*pfn_Broadcast)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t);
typedef int (*pfn_CommDestroy)(rcclComm_t);
typedef const char* (*pfn_GetErrorString)(int);
constexpr int RCCL_SUM = 0, RCCL_INT8 = 0, RCCL_F32 = 7, RCCL_BF16 = 9;      // ncclRedOp_t / ncclDataType_t values (rccl.h)

struct Rccl {
  void* lib = nullptr;
  pfn_GetUniqueId GetUniqueId = nullptr;
  pfn_CommInitRank CommInitRank = nullptr;
  pfn_AllReduce AllReduce = nullptr;
  pfn_Broadcast Broadcast = nullptr;
  pfn_CommDestroy CommDestroy = nullptr;
  pfn_GetErrorString GetErrorString = nullptr;
  char why[256] = "";
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);     // the copy already in the process
    for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) { snprintf(r.why, sizeof(r.why), "cannot load librccl.so: %s", dlerror()); return; }
    r.GetUniqueId = (pfn_GetUniqueId)dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (pfn_CommInitRank)dlsym(r.lib, "ncclCommInitRank");
    r.AllReduce = (pfn_AllReduce)dlsym(r.lib, "ncclAllReduce");
    r.Broadcast = (pfn_Broadcast)dlsym(r.lib, "ncclBroadcast");
    r.CommDestroy = (pfn_CommDestroy)dlsym(r.lib, "ncclCommDestroy");
    r.GetErrorString = (pfn_GetErrorString)dlsym(r.lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.Broadcast || !r.CommDestroy) {
      snprintf(r.why, sizeof(r.why), "librccl.so lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclBroadcast / ncclCommDestroy");
      r.lib = nullptr;
    }
  });
  return &r;
}

struct Comm { rcclComm_t comm; int rank, world; };

int dtype_of(int32_t dtype, int* out) {
  if (dtype == MG_COMM_F32) { *out = RCCL_F32; return 0; }
  if (dtype == MG_COMM_BF16) { *out = RCCL_BF16; return 0; }
  if (dtype == MG_COMM_BYTES) { *out = RCCL_INT8; return 0; }
  return -1;
}
#define MG_RCCL(call, who)                                                                                     \
  do {                                                                                                         \
    const int rc__ = (call);                                                                                   \
    if (rc__ != 0) MG_FAIL(MG_ERR_COMM, "%s: RCCL error %d (%s)", who, rc__, R->GetErrorString ? R->GetErrorString(rc__) : "?"); \
  } while (0)
}  // namespace

extern "C" int mg_comm_unique_id(uint8_t* id128) {
  Rccl* R = rccl();
  if (!R->lib) MG_FAIL(MG_ERR_COMM, "mg_comm_unique_id: %s", R->why);
  if (!id128) MG_FAIL(MG_ERR_SHAPE, "mg_comm_unique_id: null pointer");
  rcclUniqueId id;
  MG_RCCL(R->GetUniqueId(&id), "mg_comm_unique_id");
  memcpy(id128, id.internal, 128);
  return MG_OK;
}

extern "C" int mg_comm_init(void** comm_out, const uint8_t* id128, int32_t rank, int32_t world) {
  Rccl* R = rccl();
  if (!R->lib) MG_FAIL(MG_ERR_COMM, "mg_comm_init: %s", R->why);
  if (!comm_out || !id128 || world <= 0 || rank < 0 || rank >= world) MG_FAIL(MG_ERR_SHAPE, "mg_comm_init: need 0 <= rank < world and non-null pointers");
  rcclUniqueId id;
  memcpy(id.internal, id128, 128);
  Comm* c = new Comm{nullptr, rank, world};
  const int rc = R->CommInitRank(&c->comm, world, id, rank);       // binds the CURRENT HIP device
  if (rc != 0) { delete c; MG_FAIL(MG_ERR_COMM, "mg_comm_init: ncclCommInitRank failed with %d (%s)", rc, R->GetErrorString ? R->GetErrorString(rc) : "?"); }
  *comm_out = c;
  return MG_OK;
}

extern "C" int mg_comm_allreduce_sum(void* comm, void* buf, int64_t count, int32_t dtype, void* stream) {
  Rccl* R = rccl();
  Comm* c = (Comm*)comm;
  int dt;
  if (!c || !buf || count <= 0 || dtype_of(dtype, &dt) || dtype == MG_COMM_BYTES) MG_FAIL(MG_ERR_SHAPE, "mg_comm_allreduce_sum: bad communicator / buffer / count / dtype");
  MG_RCCL(R->AllReduce(buf, buf, (size_t)count, dt, RCCL_SUM, c->comm, (hipStream_t)stream), "mg_comm_allreduce_sum");
  return MG_OK;
}

extern "C" int mg_comm_broadcast(void* comm, void* buf, int64_t count, int32_t dtype, int32_t root, void* stream) {
  Rccl* R = rccl();
  Comm* c = (Comm*)comm;
  int dt;
  if (!c || !buf || count <= 0 || dtype_of(dtype, &dt) || root < 0 || root >= c->world) MG_FAIL(MG_ERR_SHAPE, "mg_comm_broadcast: bad communicator / buffer / count / dtype / root");
  MG_RCCL(R->Broadcast(buf, buf, (size_t)count, dt, root, c->comm, (hipStream_t)stream), "mg_comm_broadcast");
  return MG_OK;
}

extern "C" int mg_comm_destroy(void* comm) {
  Rccl* R = rccl();
  Comm* c = (Comm*)comm;
  if (!c) return MG_OK;
  const int rc = R->lib ? R->CommDestroy(c->comm) : 0;
  delete c;
  if (rc != 0) MG_FAIL(MG_ERR_COMM, "mg_comm_destroy: ncclCommDestroy failed with %d", rc);
  return MG_OK;
}

// ---- a compute stream that leaves CUs to the exchange -------------------------------------------------------------------
// The tile GEMMs fill every CU (one 512-thread workgroup with 128 KiB of LDS and the whole register file each), so an RCCL
// kernel enqueued on the exchange stream can only start on a CU at a GEMM workgroup boundary (~100 us), and keeps losing it.
// A stream created with a CU mask never dispatches to the masked-out CUs: run the training step's compute on it and the
// exchange always finds `reserve` free CUs (spread evenly over the XCDs).  Default off (MAGMA_DP_RESERVE_CUS, train_engine.py).
extern "C" int mg_stream_create_cu_mask(void** stream_out, int32_t reserve) {
  if (!stream_out || reserve < 0) MG_FAIL(MG_ERR_SHAPE, "mg_stream_create_cu_mask: need a non-null pointer and reserve >= 0");
  int dev = 0;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) MG_FAIL(MG_ERR_HIP, "mg_stream_create_cu_mask: %s", hipGetErrorString(e));
  const int ncu = prop.multiProcessorCount;
  if (reserve >= ncu) MG_FAIL(MG_ERR_SHAPE, "mg_stream_create_cu_mask: cannot reserve %d of %d CUs", reserve, ncu);
  const int words = (ncu + 31) / 32;
  uint32_t mask[64] = {0};
  if (words > 64) MG_FAIL(MG_ERR_UNSUPPORTED, "mg_stream_create_cu_mask: %d CUs", ncu);
  for (int i = 0; i < ncu; ++i) mask[i >> 5] |= 1u << (i & 31);
  for (int j = 0; j < reserve; ++j) {            // every (ncu / reserve)-th CU
    const int i = (int)(((int64_t)j * ncu) / reserve);
    mask[i >> 5] &= ~(1u << (i & 31));
  }
  hipStream_t st = nullptr;
  e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask);
  if (e != hipSuccess) MG_FAIL(MG_ERR_HIP, "mg_stream_create_cu_mask: hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
  *stream_out = (void*)st;
  return MG_OK;
}

extern "C" int mg_stream_destroy(void* stream) {
  if (!stream) return MG_OK;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) MG_FAIL(MG_ERR_HIP, "mg_stream_destroy: %s", hipGetErrorString(e));
  return MG_OK;
}
