// libtrhip_comm.so: the RCCL exchange behind include/trhip_comm.h (one process per GPU).  Replaces the pinned-host bounce of
// src/device_transfer.cc:140-290 and the per-pair semaphores of src/rt_renderer.cc:356-408.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

#include "../../include/trhip_comm.h"

static_assert(sizeof(ncclUniqueId) == TRHIP_COMM_ID_BYTES, "trhip_comm.h: id size");

struct trhip_comm {
    ncclComm_t comm = nullptr;
    int device = 0, nranks = 1, rank = 0;
};

std::string& trhip_comm_error_slot() {      // one message slot for the RCCL exchange here and the copy-engine exchange of comm_ipc.hip
    thread_local std::string e;
    return e;
}
namespace {
int fail(const std::string& m) { trhip_comm_error_slot() = m; return 1; }
}  // namespace

#define NCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail(std::string(#x) + ": " + ncclGetErrorString(r_)); } while (0)
#define HCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

extern "C" {

const char* trhip_comm_last_error(void) { return trhip_comm_error_slot().c_str(); }

int trhip_comm_unique_id(void* id_out) {
    if (!id_out) return fail("trhip_comm_unique_id: null out");
    NCHK(ncclGetUniqueId(static_cast<ncclUniqueId*>(id_out)));
    return 0;
}

int trhip_comm_create(int hip_device, int nranks, int rank, const void* id, trhip_comm** out) {
    if (!id || !out) return fail("trhip_comm_create: null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("trhip_comm_create: rank out of range");
    HCHK(hipSetDevice(hip_device));
    trhip_comm* c = new trhip_comm();
    c->device = hip_device; c->nranks = nranks; c->rank = rank;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, uid, rank);
    if (r != ncclSuccess) { delete c; return fail(std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
    *out = c;
    return 0;
}

void trhip_comm_destroy(trhip_comm* comm) {
    if (!comm) return;
    (void)hipSetDevice(comm->device);
    if (comm->comm) (void)ncclCommDestroy(comm->comm);
    delete comm;
}

int trhip_comm_rank(const trhip_comm* comm) { return comm ? comm->rank : -1; }
int trhip_comm_size(const trhip_comm* comm) { return comm ? comm->nranks : 0; }
int trhip_comm_get_info(const trhip_comm* comm, trhip_comm_info* out) {
    if (!comm || !out) return fail("trhip_comm_get_info: null argument");
    memset(out, 0, sizeof(*out));
    out->struct_size = (uint32_t)sizeof(*out);
    // asked of the communicator, not echoed from the arguments it was created with
    NCHK(ncclCommCount(comm->comm, &out->nranks));
    NCHK(ncclCommUserRank(comm->comm, &out->rank));
    NCHK(ncclCommCuDevice(comm->comm, &out->hip_device));
    NCHK(ncclGetVersion(&out->rccl_version));
    return 0;
}

int trhip_gather_partials(trhip_comm* comm, int root, const void* send_dev, size_t send_bytes, void* const* recv_dev, const size_t* recv_bytes,
                          void* stream) {
    if (!comm) return fail("trhip_gather_partials: null comm");
    if (root < 0 || root >= comm->nranks) return fail("trhip_gather_partials: root out of range");
    if (comm->nranks == 1) return 0;
    HCHK(hipSetDevice(comm->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (comm->rank != root) {
        if (send_bytes == 0) return 0;
        if (!send_dev) return fail("trhip_gather_partials: null send buffer");
        NCHK(ncclSend(send_dev, send_bytes, ncclUint8, root, comm->comm, s));
        return 0;
    }
    if (!recv_dev || !recv_bytes) return fail("trhip_gather_partials: the root needs the receive arrays");
    NCHK(ncclGroupStart());
    for (int r = 0; r < comm->nranks; ++r) {
        if (r == root || recv_bytes[r] == 0) continue;
        if (!recv_dev[r]) { (void)ncclGroupEnd(); return fail("trhip_gather_partials: null receive buffer"); }
        ncclResult_t e = ncclRecv(recv_dev[r], recv_bytes[r], ncclUint8, r, comm->comm, s);
        if (e != ncclSuccess) { (void)ncclGroupEnd(); return fail(std::string("ncclRecv: ") + ncclGetErrorString(e)); }
    }
    NCHK(ncclGroupEnd());
    return 0;
}

int trhip_reduce_samples(trhip_comm* comm, int root, const void* send_dev, void* recv_dev, size_t float_count, void* stream) {
    if (!comm) return fail("trhip_reduce_samples: null comm");
    if (root < 0 || root >= comm->nranks) return fail("trhip_reduce_samples: root out of range");
    if (float_count == 0) return 0;
    if (!send_dev || (comm->rank == root && !recv_dev)) return fail("trhip_reduce_samples: null buffer");
    HCHK(hipSetDevice(comm->device));
    NCHK(ncclReduce(send_dev, recv_dev, float_count, ncclFloat32, ncclSum, root, comm->comm, static_cast<hipStream_t>(stream)));
    return 0;
}

}  // extern "C"
