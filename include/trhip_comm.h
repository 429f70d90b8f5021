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
/* What RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion) - the
 * arguments of trhip_comm_create are not echoed: a job's log can show that the library saw N ranks on N devices. */
typedef struct trhip_comm_info {
    uint32_t struct_size;
    int32_t nranks, rank, hip_device;
    int32_t rccl_version;              /* NCCL_VERSION_CODE of the linked librccl, e.g. 22204 */
} trhip_comm_info;
int trhip_comm_get_info(const trhip_comm* comm, trhip_comm_info* out);
/* device_transfer of one frame: every rank but `root` sends `send_bytes` bytes at `send_dev` (its partial colour target),
 * `root` receives recv_bytes[r] bytes into recv_dev[r] for every r != root (entries of index root are ignored; the arrays
 * may be NULL on the other ranks).  Sizes may differ from rank to rank (shuffled strips with balanced shares,
 * src/distribution_strategy.cc:62-69); a size of zero skips that peer on both sides.  One ncclGroupStart / ncclGroupEnd. */
int trhip_gather_partials(trhip_comm* comm, int root, const void* send_dev, size_t send_bytes, void* const* recv_dev,
                          const size_t* recv_bytes, void* stream);
/* Sample shards: the sum over all ranks of `float_count` floats at send_dev lands in recv_dev on
 * `root` (recv_dev may equal send_dev; it is ignored on the other ranks).  ncclReduce(sum). */
int trhip_reduce_samples(trhip_comm* comm, int root, const void* send_dev, void* recv_dev, size_t float_count, void* stream);


/* ---- The same gather on the copy engines: no kernel of a library runs on either side.
 *
 * trhip_gather_partials is an RCCL send / receive pair per peer: RCCL moves the bytes with kernels of its own, which need CUs on
 * both devices - on the display rank, next to persistent trace kernels that fill the chip.  trhip_ipc moves them the way the
 * single-process host does (trhip_copy_peer = hipMemcpyPeerAsync, tauray_hip.hh) and the way the reference's device_transfer
 * does in spirit (src/device_transfer.cc:140-290: plain copies paced by semaphores), across process boundaries: the display rank
 * owns one receive arena (slots x peers x slot_bytes of device memory) and exports it with hipIpcGetMemHandle; every other rank
 * maps it (hipIpcOpenMemHandle) and writes its partial frame there with hipMemcpyAsync on its own stream - a DMA over its xGMI
 * link to the display device, no CU involved.  Ordering travels in 8-byte tags: behind its copy a sender writes the frame's
 * number into the display rank's tag block, the display rank's stream waits for the tags of all peers (one wave of a one-block
 * kernel that polls, the only device code of the exchange) before whatever consumes the frames; when the consumer is enqueued the
 * display rank releases the slot by writing a tag into every sender's block, which a sender's stream waits for before it
 * overwrites that slot `slots` frames later.  Nothing synchronises with the host.
 *
 * Set-up: every rank creates its end, exports TRHIP_IPC_EXPORT_BYTES bytes, the caller gathers the blobs of all ranks by its own
 * means (as with the communicator id) and hands them, rank-major, to trhip_ipc_connect.  Ranks may share a device (the two-process
 * test on one GPU), not a process (an IPC handle cannot be opened by the process that made it). */
#define TRHIP_IPC_EXPORT_BYTES 256
typedef struct trhip_ipc trhip_ipc;
/* slot_bytes: the largest partial frame a peer will send (get_distribution_target_max_size x layers x 16); slots: frames that may
 * be in flight between a sender and the display rank (the renderer's frame slots). */
int trhip_ipc_create(int hip_device, int nranks, int rank, int root, size_t slot_bytes, int slots, trhip_ipc** out);
int trhip_ipc_export(trhip_ipc* ipc, void* blob_out);
int trhip_ipc_connect(trhip_ipc* ipc, const void* blobs_of_all_ranks);
/* One frame.  Not the root: waits (on `stream`) until the slot of this frame has been released, copies send_bytes bytes from
 * send_dev into it, posts the arrival tag.  Root: waits (on `stream`) for the arrival tags of every peer r with recv_bytes[r] > 0
 * and returns where their partial frames are in recv_dev_out[r] (pointers into the arena, valid until trhip_ipc_release).
 * A device-side wait gives up after 10 s (TRHIP_IPC_TIMEOUT_MS) so that a dead peer cannot hang the device; the next call into the
 * exchange then fails ("... gave up"): the frame behind that wait is incomplete. */
int trhip_ipc_gather_partials(trhip_ipc* ipc, const void* send_dev, size_t send_bytes, void** recv_dev_out, const size_t* recv_bytes, void* stream);
/* Root, after the consumers of the frame just gathered have been enqueued on `stream`: the slot may be overwritten once they are done. */
int trhip_ipc_release(trhip_ipc* ipc, void* stream);
/* Non-zero (and the reason in trhip_comm_last_error) once a device-side wait has given up.  Every call above begins with this check; a
 * caller that is about to use a frame - write it out, display it - calls it after synchronising the frame's stream, because the last
 * frame of a job is followed by no other call (tr::process_rt_renderer::finish_frame, tauray_amd.comm.IpcExchange do).
 * Contract of the calls above, for the record: the wait kernel is one wave that polls with s_sleep between system-scope loads - it holds
 * one wave slot of the device for as long as a peer is late, at most the timeout; the host never blocks in a gather or a release unless it
 * runs more than 1 024 of them ahead of its device (their tag values wait in a pinned ring), in which case the call waits for the oldest;
 * a failed call leaves the frame counter where it was. */
int trhip_ipc_check(trhip_ipc* ipc);
void trhip_ipc_destroy(trhip_ipc* ipc);

#ifdef __cplusplus
}
#endif
#endif
