/* mmvqa_comm.h -- thin RCCL wrappers of the data-parallel exchange (libmmvqa_comm.so).
 *
 * SURVEY.md 8(b) "Minimum C-ABI surface": `mmvqa_allreduce_bucket(handle, stream, ptr, n)` / `mmvqa_allgather(...)`
 * "thin wrappers over RCCL comms created by the Python launcher".  The reference trains single-process
 * (pretrain/roco_train.py:74); this is the native side of the data-parallel layer the build adds:
 *   collective 1: all-reduce (sum) of the flat fp32 gradient buffer in buckets, in place, on the caller's stream;
 *   collective 2: all-gather of the [2N, 128] SupCon features (models/SupConLoss/supcon_utils.py:283-287 over the global
 *                 view set).
 * A separate library so that libmmvqa_hip.so carries no RCCL dependency; `mmvqa_amd.ddp.NativeComm` binds it.  The
 * default exchange of bench.py / train.py stays torch.distributed (backend "nccl" = RCCL), which needs no second
 * communicator; `--native-comm` switches the gradient all-reduce to these entry points.
 * Every call: borrowed device pointers, the caller's hipStream_t, asynchronous, 0 on success / negative code with
 * mmvqa_comm_last_error(). */
#ifndef MMVQA_COMM_H
#define MMVQA_COMM_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mmvqa_comm mmvqa_comm;
#define MMVQA_COMM_ID_BYTES 128

const char* mmvqa_comm_last_error(void);
/* rank 0 draws the rendezvous id (ncclGetUniqueId) and hands its bytes to the other ranks (any side channel: the
 * launcher broadcasts them over torch.distributed's store / gloo) */
int mmvqa_comm_unique_id(void* id_out /* MMVQA_COMM_ID_BYTES */);
/* collective over all ranks: one communicator per process on the CURRENT device */
int mmvqa_comm_create(const void* id, int rank, int world, mmvqa_comm** out);
int mmvqa_comm_destroy(mmvqa_comm* c);
int mmvqa_comm_rank(const mmvqa_comm* c);
int mmvqa_comm_world(const mmvqa_comm* c);
int mmvqa_comm_rccl_version(void);
/* in-place sum over ranks of n floats (one bucket of the flat gradient buffer), ordered on `stream` */
int mmvqa_allreduce_bucket(mmvqa_comm* c, void* stream, float* ptr, long long n);
/* recv[r * n .. (r+1) * n) = send of rank r */
int mmvqa_allgather(mmvqa_comm* c, void* stream, const float* send, float* recv, long long n);
/* rank `root`'s n floats to every rank (start-up: identical replicas) */
int mmvqa_broadcast(mmvqa_comm* c, void* stream, float* ptr, long long n, int root);

#ifdef __cplusplus
}
#endif
#endif
