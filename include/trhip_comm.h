/* trhip_comm - the inter-device exchange of the path-tracing core for one process per GPU: RCCL over xGMI behind a C ABI.
 *
 * Replaces tr::device_transfer (src/device_transfer.cc:21-347: GPU -> pinned host memory -> GPU copies paced by exported
 * timeline semaphores, created per device pair by src/rt_renderer.cc:356-408) for hosts that run one process per GPU:
 * the partial frames of a pixel-sharded frame travel to the display rank in one grouped ncclSend / ncclRecv exchange, each
 * peer's slab over its own xGMI link, straight from and into device memory on the caller's stream; sample shards are summed
 * with ncclReduce.  (A host that drives all devices from one process, like the reference, uses trhip_copy_peer of trhip.h.)
 *
 * libtrhip_comm.so is a library of its own, linked against librccl: libtrhip.so does not depend on it, so a single-GPU
 * user never maps RCCL.  Conventions as in trhip.h: 0 = success, otherwise trhip_comm_last_error() has the message; calls are
 * asynchronous on `stream` (a hipStream_t passed as void*, NULL = the default stream) unless stated.
 */
#ifndef TRHIP_COMM_H
#define TRHIP_COMM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRHIP_COMM_ID_BYTES 128          /* sizeof(ncclUniqueId) */

typedef struct trhip_comm trhip_comm;

const char* trhip_comm_last_error(void);
/* ncclGetUniqueId: called by one rank (the display rank), whose caller hands the 128 bytes to every other rank by its own
 * means - a file, an environment variable, MPI, a torch.distributed store. */
int trhip_comm_unique_id(void* id_out);
/* ncclCommInitRank on HIP device `hip_device`: collective over the `nranks` processes that hold the same id; blocks until
 * all of them have called it. */
int trhip_comm_create(int hip_device, int nranks, int rank, const void* id, trhip_comm** out);
void trhip_comm_destroy(trhip_comm* comm);
int trhip_comm_rank(const trhip_comm* comm);
int trhip_comm_size(const trhip_comm* comm);
/* device_transfer of one frame: every rank but `root` sends `send_bytes` bytes at `send_dev` (its partial colour target),
 * `root` receives recv_bytes[r] bytes into recv_dev[r] for every r != root (entries of index root are ignored; the arrays
 * may be NULL on the other ranks).  Sizes may differ from rank to rank (shuffled strips with balanced shares,
 * src/distribution_strategy.cc:62-69); a size of zero skips that peer on both sides.  One ncclGroupStart / ncclGroupEnd. */
int trhip_gather_partials(trhip_comm* comm, int root, const void* send_dev, size_t send_bytes, void* const* recv_dev,
                          const size_t* recv_bytes, void* stream);
/* Sample shards: the sum over all ranks of `float_count` floats at send_dev lands in recv_dev on
 * `root` (recv_dev may equal send_dev; it is ignored on the other ranks).  ncclReduce(sum). */
int trhip_reduce_samples(trhip_comm* comm, int root, const void* send_dev, void* recv_dev, size_t float_count, void* stream);

#ifdef __cplusplus
}
#endif
#endif
