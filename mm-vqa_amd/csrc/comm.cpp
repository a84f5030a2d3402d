// Thin RCCL wrappers of the data-parallel exchange (include/mmvqa_comm.h): libmmvqa_comm.so.
#include <rccl/rccl.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/mmvqa_comm.h"

static_assert(sizeof(ncclUniqueId) == MMVQA_COMM_ID_BYTES, "ncclUniqueId size");

struct mmvqa_comm {
  ncclComm_t comm;
  int rank, world;
};

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define NCCL_TRY(x)                                                                          \
  do {                                                                                       \
    ncclResult_t r_ = (x);                                                                   \
    if (r_ != ncclSuccess) return fail(-2, "%s failed: %s", #x, ncclGetErrorString(r_));    \
  } while (0)

extern "C" {

const char* mmvqa_comm_last_error(void) { return g_err; }

int mmvqa_comm_unique_id(void* id_out) {
  if (!id_out) return fail(-1, "comm_unique_id: null pointer");
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

int mmvqa_comm_create(const void* id, int rank, int world, mmvqa_comm** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(-1, "comm_create: bad rank %d / world %d", rank, world);
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  mmvqa_comm* c = new mmvqa_comm{nullptr, rank, world};
  ncclResult_t r = ncclCommInitRank(&c->comm, world, uid, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(-2, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
  }
  *out = c;
  return 0;
}

int mmvqa_comm_destroy(mmvqa_comm* c) {
  if (!c) return 0;
  ncclResult_t r = ncclCommDestroy(c->comm);
  delete c;
  return r == ncclSuccess ? 0 : fail(-2, "ncclCommDestroy failed: %s", ncclGetErrorString(r));
}

int mmvqa_comm_rank(const mmvqa_comm* c) { return c ? c->rank : -1; }
int mmvqa_comm_world(const mmvqa_comm* c) { return c ? c->world : 0; }
int mmvqa_comm_rccl_version(void) {
  int v = 0;
  return ncclGetVersion(&v) == ncclSuccess ? v : -1;
}

int mmvqa_allreduce_bucket(mmvqa_comm* c, void* stream, float* ptr, long long n) {
  if (!c || !ptr || n < 0) return fail(-1, "allreduce_bucket: bad argument");
  if (n == 0) return 0;
  NCCL_TRY(ncclAllReduce(ptr, ptr, (size_t)n, ncclFloat, ncclSum, c->comm, (hipStream_t)stream));
  return 0;
}

int mmvqa_allgather(mmvqa_comm* c, void* stream, const float* send, float* recv, long long n) {
  if (!c || !send || !recv || n < 0) return fail(-1, "allgather: bad argument");
  if (n == 0) return 0;
  NCCL_TRY(ncclAllGather(send, recv, (size_t)n, ncclFloat, c->comm, (hipStream_t)stream));
  return 0;
}

int mmvqa_broadcast(mmvqa_comm* c, void* stream, float* ptr, long long n, int root) {
  if (!c || !ptr || n < 0 || root < 0 || root >= c->world) return fail(-1, "broadcast: bad argument");
  if (n == 0) return 0;
  NCCL_TRY(ncclBroadcast(ptr, ptr, (size_t)n, ncclFloat, root, c->comm, (hipStream_t)stream));
  return 0;
}

}  // extern "C"
