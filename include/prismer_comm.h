/* prismer_comm.h -- C ABI of libprismer_comm.so: the gradient-exchange transport of the data-parallel Prismer training
 * step (one process per MI355X, RCCL over xGMI).
 *
 * Replaces: the DistributedDataParallel gradient all-reduce that accelerate wraps around the model
 * (train_caption.py:92-93 `accelerator = Accelerator(...)`, :117 `accelerator.prepare(model, ...)`, :132
 * `accelerator.backward(loss)`; the same lines in train_vqa.py) -- SURVEY.md section 8(b) "RCCL side".
 *
 * Conventions (same as prismer_hip.h): plain C, raw device pointers, every call only ENQUEUES on `stream`, no device
 * allocation, return PH_COMM_OK (0) or a negative code; ph_comm_last_error() returns a thread-local message.
 * RCCL is resolved at run time (dlopen of the librccl the process already holds -- torch ships one -- else the system one),
 * so the library itself has no link-time dependency on it.
 */
#ifndef PRISMER_COMM_H_
#define PRISMER_COMM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
#ifndef PRISMER_HIP_H_
typedef struct ihipStream_t* hipStream_t;
#endif
#endif

enum { PH_COMM_OK = 0, PH_COMM_ERR_BAD_ARG = -1, PH_COMM_ERR_NO_RCCL = -2, PH_COMM_ERR_RCCL = -3 };
enum { PH_COMM_F32 = 0, PH_COMM_BF16 = 1 };
#define PH_COMM_UNIQUE_ID_BYTES 128

typedef struct ph_comm ph_comm;   /* opaque communicator handle */

/* rank 0 creates the rendezvous token (ncclGetUniqueId) and ships its 128 bytes to the other ranks by any side channel
 * (the Python host broadcasts it over the process group's store). */
int ph_comm_unique_id(void* out_id_128);
/* collective over all ranks: every rank calls it with the same token; the current HIP device is the rank's GPU */
int ph_comm_init(int rank, int world, const void* unique_id_128, ph_comm** out);
/* in-place SUM all-reduce of `count` elements of `dtype` (one gradient bucket), enqueued on `stream` */
int ph_allreduce_bucket(ph_comm* comm, void* buf, int64_t count, int dtype, hipStream_t stream);
/* The reduce-scatter + all-gather form of the exchange (round 6; the communication pattern of accelerate's FSDP SHARD_GRAD_OP flag,
 * train_caption.py:56-66, Trainer(shard_optimizer='rs_ag')): rank r receives the SUM of chunk r of `world` equal chunks of `send`
 * (`world * recv_count` elements); the all-gather is its inverse (`recv` holds `world * send_count` elements); ph_broadcast ships `count`
 * elements of `buf` from rank `root` in place (the owners' updated slices of the ZeRO-1 style sharded optimizer, train_pretrain.py:56-91). */
int ph_reduce_scatter(ph_comm* comm, const void* send, void* recv, int64_t recv_count, int dtype, hipStream_t stream);
int ph_all_gather(ph_comm* comm, const void* send, void* recv, int64_t send_count, int dtype, hipStream_t stream);
int ph_broadcast(ph_comm* comm, void* buf, int64_t count, int dtype, int root, hipStream_t stream);
int ph_comm_world(const ph_comm* comm);
int ph_comm_destroy(ph_comm* comm);
const char* ph_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PRISMER_COMM_H_ */
