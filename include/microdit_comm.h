/* C ABI of libmicrodit_comm.so — the data-parallel gradient exchange of the MicroDiT training path straight on RCCL (xGMI),
 * without a torch.distributed process group on the data path (md_comm_{init, allreduce_bucket, destroy}).
 *
 * Replaces, for this path: Composer's FSDP gradient reduction / parameter all-gather (reference configs/res_256_pretrain.yaml:
 * 117-118 `fsdp_config: sharding_strategy: SHARD_GRAD_OP`, micro_diffusion/models/model.py:100-102 wrapping) and the NCCL calls
 * torch issues under it.  One communicator per process (= per GPU); every collective is asynchronous:
 *
 *   - it is enqueued on the communicator's OWN high-priority HIP stream, ordered behind everything already enqueued on
 *     `order_after` (the compute stream that produced the bucket) by an event, so it overlaps the kernels the caller enqueues next;
 *   - it returns a ticket; md_comm_wait(comm, ticket, stream) makes `stream` wait for that collective (device-side dependency,
 *     the host never blocks).
 *
 * Bootstrap: rank 0 calls md_comm_unique_id and hands the 128 bytes to the other ranks by any side channel (the launcher's
 * store, a file, torch.distributed's object broadcast); every rank then calls md_comm_init with the same bytes.
 * Every function returns 0, MD_COMM_BAD_ARG (-1), MD_COMM_NO_RCCL (-2: librccl.so not loadable) or MD_COMM_FAILED (-3: an RCCL /
 * HIP call failed; md_comm_last_error() has the text).  No function allocates device memory. */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MD_COMM_ABI_VERSION 2
#define MD_COMM_UNIQUE_ID_BYTES 128
#define MD_COMM_BAD_ARG (-1)
#define MD_COMM_NO_RCCL (-2)
#define MD_COMM_FAILED (-3)

enum md_comm_dtype { MD_COMM_BF16 = 0, MD_COMM_F32 = 1 };

typedef struct md_comm md_comm;
typedef void* md_comm_stream;   /* a hipStream_t */

int md_comm_abi_version(void);
const char* md_comm_last_error(void);
int md_comm_unique_id(void* out_128_bytes);
/* device: HIP device ordinal this rank computes on (the communicator and its stream are created there). */
int md_comm_init(md_comm** out, const void* unique_id_128_bytes, int32_t rank, int32_t world, int32_t device);
int md_comm_destroy(md_comm* c);
int md_comm_rank(const md_comm* c);
int md_comm_world(const md_comm* c);

/* buf[i] = sum over ranks (in place).  The all-reduce form of the gradient exchange (trainer.GradSync mode "allreduce") and the
 * one-dimensional "small" bucket / the scalar norm of the sharded form. */
int md_comm_allreduce_bucket(md_comm* c, void* buf, int64_t count, int32_t dtype, md_comm_stream order_after, int64_t* ticket);
/* recv[i] = sum over ranks of send[rank * recv_count + i]: rank r receives the sum of chunk r of a bucket of world * recv_count
 * elements (FSDP SHARD_GRAD_OP's gradient reduction). */
int md_comm_reduce_scatter_bucket(md_comm* c, const void* send, void* recv, int64_t recv_count, int32_t dtype,
                                  md_comm_stream order_after, int64_t* ticket);
/* recv[r * send_count + i] = rank r's send[i]: the fresh bf16 weights of every rank's chunk, gathered under the next forward. */
int md_comm_all_gather_bucket(md_comm* c, const void* send, void* recv, int64_t send_count, int32_t dtype,
                              md_comm_stream order_after, int64_t* ticket);
/* `stream` waits (on the device) for the collective that returned `ticket`; ticket <= 0 is a no-op.  Tickets are valid for the
 * MD_COMM_TICKETS (1024) most recent collectives of the communicator. */
int md_comm_wait(md_comm* c, int64_t ticket, md_comm_stream stream);
/* 1 = the collective that returned `ticket` has finished on the device, 0 = not yet (host-side poll, never blocks); < 0 = error.
 * The step uses it to hand RCCL's CUs back to the GEMM grids as soon as the wire is idle (trainer.GradSync.in_flight). */
int md_comm_query(md_comm* c, int64_t ticket);
/* The host blocks until every collective issued so far has finished (tests, shutdown). */
int md_comm_synchronize(md_comm* c);

#ifdef __cplusplus
}
#endif
