// C-ABI wrappers over librccl.so for the collectives of the train steps: gradient all-reduce (mean / sum /
// max), parameter broadcast and the DLRM all-to-all with per-peer split sizes, each enqueued on the HIP stream the caller names
// (the engines' communication stream, event-fenced against the compute stream on the Python side).
//
// Replaces what the reference reaches through torch.distributed's ProcessGroupNCCL:
//   Classification/ConvNets/image_classification/training.py:78-84 (DDP reducer), LanguageModeling/BERT/run_pretraining.py:461-470
//   (the comm hook's all_reduce), Recommendation/DLRM/dlrm/model/distributed.py:68,95 (all_to_all).
// RCCL is bound with dlopen at first use -- the library that torch already mapped when there is one (one RCCL instance per
// process), /opt/rocm/lib/librccl.so.1 otherwise -- so libdle_mi355x.so itself carries no link-time dependency on it and the
// single-GPU paths never touch it.  The unique id travels over whatever rendezvous the caller has (utils/rccl.py: the
// MASTER_ADDR / MASTER_PORT store of torch.distributed).
#include "common.h"
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

typedef struct { char internal[128]; } dle_ncclUniqueId;      // NCCL_UNIQUE_ID_BYTES = 128 (rccl.h:40-43)
typedef void* dle_ncclComm_t;
typedef int (*fn_GetUniqueId)(dle_ncclUniqueId*);
typedef int (*fn_CommInitRank)(dle_ncclComm_t*, int, dle_ncclUniqueId, int);
typedef int (*fn_CommDestroy)(dle_ncclComm_t);
typedef int (*fn_CommCount)(const dle_ncclComm_t, int*);
typedef int (*fn_AllReduce)(const void*, void*, size_t, int, int, dle_ncclComm_t, hipStream_t);
typedef int (*fn_Broadcast)(const void*, void*, size_t, int, int, dle_ncclComm_t, hipStream_t);
typedef int (*fn_SendRecv)(void*, size_t, int, int, dle_ncclComm_t, hipStream_t);
typedef int (*fn_Group)(void);
typedef const char* (*fn_ErrorString)(int);

static struct {
  void* h;
  fn_GetUniqueId GetUniqueId; fn_CommInitRank CommInitRank; fn_CommDestroy CommDestroy; fn_CommCount CommCount;
  fn_AllReduce AllReduce; fn_Broadcast Broadcast; fn_SendRecv Send; fn_SendRecv Recv; fn_Group GroupStart; fn_Group GroupEnd;
  fn_ErrorString GetErrorString;
  int tried;
} R;

static int rccl_load() {
  if (R.tried) return R.h != nullptr;
  R.tried = 1;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (int pass = 0; pass < 2 && !R.h; ++pass)
    for (const char* n : names) {
      R.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));       // pass 0: the copy torch has mapped
      if (R.h) break;
    }
  if (!R.h) { dle_set_error("rccl: librccl.so not found (%s)", dlerror()); return 0; }
#define SYM(field, name) do { R.field = (decltype(R.field))dlsym(R.h, name); if (!R.field) { dle_set_error("rccl: symbol %s missing", name); R.h = nullptr; return 0; } } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(CommCount, "ncclCommCount"); SYM(AllReduce, "ncclAllReduce"); SYM(Broadcast, "ncclBroadcast"); SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  return 1;
}

#define RCCL_CALL(expr, what) do { const int rc_ = (expr); if (rc_ != 0) { \
    dle_set_error("rccl %s: %s", what, R.GetErrorString ? R.GetErrorString(rc_) : "error"); return 1000 + rc_; } } while (0)

// DLE dtype -> ncclDataType_t (rccl.h:459-468); 100 / 101 / 102: int32 / int64 / uint8 (flags, row ids, raw bytes)
static int nccl_dtype(int dt, size_t* esz) {
  switch (dt) {
    case DLE_F32: *esz = 4; return 7;
    case DLE_F16: *esz = 2; return 6;
    case DLE_BF16: *esz = 2; return 9;
    case 100: *esz = 4; return 2;
    case 101: *esz = 8; return 4;
    case 102: *esz = 1; return 1;
    default: *esz = 0; return -1;
  }
}

// 1 when librccl.so could be bound in this process.
extern "C" int dle_rccl_available(void) { return rccl_load(); }

// Rank 0: a fresh 128-byte unique id for one communicator (ncclGetUniqueId); the caller hands it to every rank.
extern "C" int dle_rccl_unique_id(void* id128) {
  DLE_CHECK_ARG(id128, "rccl_unique_id: null pointer");
  if (!rccl_load()) return 2;
  dle_ncclUniqueId id;
  RCCL_CALL(R.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id128, &id, sizeof(id));
  return 0;
}

// ncclCommInitRank on the CURRENT device.  -> 0 and *comm_out = an opaque handle.
extern "C" int dle_rccl_init(const void* id128, int rank, int world, void** comm_out) {
  DLE_CHECK_ARG(id128 && comm_out && world >= 1 && rank >= 0 && rank < world, "rccl_init: bad arguments");
  if (!rccl_load()) return 2;
  dle_ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  dle_ncclComm_t c = nullptr;
  RCCL_CALL(R.CommInitRank(&c, world, id, rank), "ncclCommInitRank");
  *comm_out = c;
  return 0;
}

// Number of ranks of the communicator as RCCL reports it (ncclCommCount): the proof that N ranks joined.
extern "C" int dle_rccl_count(void* comm) {
  if (!comm || !rccl_load()) return -1;
  int n = -1;
  if (R.CommCount((dle_ncclComm_t)comm, &n) != 0) return -1;
  return n;
}

extern "C" int dle_rccl_destroy(void* comm) {
  if (!comm || !rccl_load()) return 0;
  RCCL_CALL(R.CommDestroy((dle_ncclComm_t)comm), "ncclCommDestroy");
  return 0;
}

// In-place all-reduce of `count` elements on `stream`; op: 0 sum, 2 max, 4 average (ncclRedOp_t, rccl.h:448-452).
extern "C" int dle_rccl_allreduce(void* comm, void* buf, int64_t count, int dtype, int op, hipStream_t stream) {
  DLE_CHECK_ARG(comm && (buf || count == 0) && count >= 0 && (op == 0 || op == 2 || op == 4), "rccl_allreduce: bad arguments");
  size_t esz;
  const int dt = nccl_dtype(dtype, &esz);
  DLE_CHECK_ARG(dt >= 0, "rccl_allreduce: unsupported dtype %d", dtype);
  if (count == 0) return 0;
  RCCL_CALL(R.AllReduce(buf, buf, (size_t)count, dt, op, (dle_ncclComm_t)comm, stream), "ncclAllReduce");
  return 0;
}

// In-place broadcast of `bytes` bytes from rank `root`.
extern "C" int dle_rccl_broadcast(void* comm, void* buf, int64_t bytes, int root, hipStream_t stream) {
  DLE_CHECK_ARG(comm && (buf || bytes == 0) && bytes >= 0, "rccl_broadcast: bad arguments");
  if (bytes == 0) return 0;
  RCCL_CALL(R.Broadcast(buf, buf, (size_t)bytes, 1 /* ncclUint8 */, root, (dle_ncclComm_t)comm, stream), "ncclBroadcast");
  return 0;
}

// All-to-all with per-peer sizes: peer p receives send_bytes[p] bytes starting at send + sum(send_bytes[:p]) and delivers
// recv_bytes[p] bytes at recv + sum(recv_bytes[:p]) (torch.distributed.all_to_all_single with split lists,
// dlrm/model/distributed.py:68,95) -- one grouped ncclSend / ncclRecv pair per peer.  The size arrays live on the HOST.
extern "C" int dle_rccl_alltoallv(void* comm, const void* send, const int64_t* send_bytes, void* recv, const int64_t* recv_bytes,
                                  int world, hipStream_t stream) {
  DLE_CHECK_ARG(comm && send_bytes && recv_bytes && world >= 1, "rccl_alltoallv: bad arguments");
  RCCL_CALL(R.GroupStart(), "ncclGroupStart");
  int64_t so = 0, ro = 0;
  for (int p = 0; p < world; ++p) {
    if (send_bytes[p] < 0 || recv_bytes[p] < 0) { R.GroupEnd(); dle_set_error("rccl_alltoallv: negative size"); return 1; }
    if (send_bytes[p] > 0) {
      const int rc = R.Send((void*)((const char*)send + so), (size_t)send_bytes[p], 1, p, (dle_ncclComm_t)comm, stream);
      if (rc != 0) { R.GroupEnd(); dle_set_error("rccl ncclSend: %s", R.GetErrorString(rc)); return 1000 + rc; }
    }
    if (recv_bytes[p] > 0) {
      const int rc = R.Recv((char*)recv + ro, (size_t)recv_bytes[p], 1, p, (dle_ncclComm_t)comm, stream);
      if (rc != 0) { R.GroupEnd(); dle_set_error("rccl ncclRecv: %s", R.GetErrorString(rc)); return 1000 + rc; }
    }
    so += send_bytes[p];
    ro += recv_bytes[p];
  }
  RCCL_CALL(R.GroupEnd(), "ncclGroupEnd");
  return 0;
}
