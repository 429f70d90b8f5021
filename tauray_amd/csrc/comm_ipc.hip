// The copy-engine exchange of include/trhip_comm.h (trhip_ipc_*): partial frames travel by hipMemcpyAsync into IPC-mapped memory
// of the display rank, ordering by 8-byte tags.  Part of libtrhip_comm.so.  Replaces src/device_transfer.cc:140-290 for one
// process per GPU when the RCCL kernels of trhip_gather_partials are not wanted on the devices.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/trhip_comm.h"

namespace {

int ipc_fail(const std::string& m);

// One lane per tag: polls until the tag has reached `want` (tags only grow).  System-scope acquire loads: the tag is written by
// another device's DMA engine (or another process's copy), and what it announces - the partial frame - was written before it.
// Gives up after ~10 s of wall clock (a peer that died; TRHIP_IPC_TIMEOUT_MS) and says so in *timed_out - pinned host memory, which the
// next call into the exchange reads and turns into an error: the frame behind a wait that gave up holds whatever was in the arena.
__global__ void k_wait_tags(const unsigned long long* tags, int n, int stride, const unsigned char* wanted, unsigned long long want, int* timed_out,
                            unsigned long long timeout_ticks) {
    const int i = threadIdx.x;
    if (i >= n || !wanted[i]) return;
    const unsigned long long* p = tags + (size_t)i * stride;
    const unsigned long long t0 = wall_clock64();       // 100 MHz
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > timeout_ticks) { __hip_atomic_store(timed_out, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); return; }
    }
}

struct Blob {      // what a rank tells the others (TRHIP_IPC_EXPORT_BYTES)
    char magic[8];
    int rank, root, nranks, slots;
    unsigned long long slot_bytes;
    int has_arena, pad;
    hipIpcMemHandle_t arena;      // root: the receive arena
    hipIpcMemHandle_t tags;       // root: arrival tags [nranks][slots]; others: release tags [slots]
};
static_assert(sizeof(Blob) <= TRHIP_IPC_EXPORT_BYTES, "trhip_comm.h: export size");

// What the asynchronous copies of a gather read from the host - the tag value and the mask of peers to wait for - waits in pinned rings
// until the copy engines get to it.  A ring entry is reused TAG_RING calls later; an event recorded behind the call's last copy says
// whether the device is done with it: a caller more than TAG_RING calls ahead of its device is held until it is (it never was: a frame
// slot holds a handful of frames, but a ring that wraps unchecked is a contract nobody can rely on).
constexpr int TAG_RING = 1024;

}  // namespace

struct trhip_ipc {
    int device = 0, nranks = 1, rank = 0, root = 0, slots = 1;
    size_t slot_bytes = 0;
    unsigned long long frame = 0;            // gathers so far
    bool connected = false;
    // own allocations
    void* arena = nullptr;                    // root
    unsigned long long* tags = nullptr;       // root: [nranks][slots] arrival tags; others: [slots] release tags
    int* timed_out = nullptr;                 // pinned host memory: set by a wait that gave up, read by the next call
    unsigned long long timeout_ticks = 1000000000ull;   // wall_clock64: 100 MHz
    unsigned char* wanted_dev = nullptr;
    unsigned long long* tag_values = nullptr; // pinned ring: sources of the tag copies (gathers of a sender; releases of the root use the upper half)
    unsigned char* wanted_ring = nullptr;     // pinned ring: 64 bytes per gather, the source of the copy into wanted_dev
    std::vector<hipEvent_t> ring_events;      // one per ring entry, recorded behind the last copy that reads it
    unsigned long long releases = 0;
    // mapped from the other side
    void* root_arena = nullptr;               // non-root
    unsigned long long* root_tags = nullptr;  // non-root
    std::vector<unsigned long long*> peer_tags;   // root: release tags of every peer
};

std::string& trhip_comm_error_slot();      // comm.cc: what trhip_comm_last_error() returns
namespace {
int ipc_fail(const std::string& m) { trhip_comm_error_slot() = m; return 1; }
#define ICHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return ipc_fail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
// ring entry k may be written again once the copies that read it last time have run
int ring_acquire(trhip_ipc* c, size_t k) {
    if (c->ring_events[k] && hipEventQuery(c->ring_events[k]) != hipSuccess) { (void)hipGetLastError(); ICHK(hipEventSynchronize(c->ring_events[k])); }
    return 0;
}
int ring_release(trhip_ipc* c, size_t k, hipStream_t s) {
    if (!c->ring_events[k]) ICHK(hipEventCreateWithFlags(&c->ring_events[k], hipEventDisableTiming));
    ICHK(hipEventRecord(c->ring_events[k], s));
    return 0;
}
}  // namespace

extern "C" {

int trhip_ipc_create(int hip_device, int nranks, int rank, int root, size_t slot_bytes, int slots, trhip_ipc** out) {
    if (!out) return ipc_fail("trhip_ipc_create: null out");
    if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks || root < 0 || root >= nranks) return ipc_fail("trhip_ipc_create: rank out of range (at most 64 ranks)");
    if (slots < 1 || slots > 16 || slot_bytes == 0) return ipc_fail("trhip_ipc_create: 1 ... 16 slots of at least one byte");
    ICHK(hipSetDevice(hip_device));
    trhip_ipc* c = new trhip_ipc();
    c->device = hip_device; c->nranks = nranks; c->rank = rank; c->root = root; c->slots = slots;
    c->slot_bytes = (slot_bytes + 255) & ~(size_t)255;
    const size_t n_tags = rank == root ? (size_t)nranks * slots : (size_t)slots;
    hipError_t e = hipSuccess;
    // What other devices write - the arena and the tags - is fine-grained device memory, as RCCL allocates the buffers its peers write:
    // coherent for writers outside this device without relying on what an L2 does with lines of ordinary (coarse-grained) allocations
    // between kernels.  (TRHIP_IPC_COARSE=1: plain hipMalloc, for A/B.)
    // No silent fall-back: on coarse-grained memory a polling load may be served by an L2 that never sees the peer's write - a ten-second
    // stall and a "gave up" with nothing pointing at the allocation.  Whoever wants plain hipMalloc says so.
    auto shared_alloc = [](void** p, size_t bytes) {
        static const bool coarse = getenv("TRHIP_IPC_COARSE") && atoi(getenv("TRHIP_IPC_COARSE")) != 0;
        return coarse ? hipMalloc(p, bytes) : hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained);
    };
    if (rank == root) e = shared_alloc(&c->arena, (size_t)nranks * slots * c->slot_bytes);
    if (e == hipSuccess) e = shared_alloc(reinterpret_cast<void**>(&c->tags), n_tags * 8);
    if (e == hipSuccess) e = hipMemset(c->tags, 0, n_tags * 8);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->timed_out), 4, hipHostMallocDefault);
    if (e == hipSuccess) *c->timed_out = 0;
    if (const char* t = getenv("TRHIP_IPC_TIMEOUT_MS")) if (atof(t) > 0) c->timeout_ticks = (unsigned long long)(atof(t) * 1e5);
    if (e == hipSuccess) e = hipMalloc(&c->wanted_dev, 64 * (size_t)slots);
    if (e == hipSuccess) e = hipHostMalloc(&c->tag_values, 2 * TAG_RING * 8, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(&c->wanted_ring, TAG_RING * 64, hipHostMallocDefault);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        const std::string what = hipGetErrorString(e);
        trhip_ipc_destroy(c);
        return ipc_fail("trhip_ipc_create: " + what + " (the arena and the tags are fine-grained device memory, hipExtMallocWithFlags; TRHIP_IPC_COARSE=1 takes hipMalloc instead)");
    }
    c->ring_events.assign(2 * TAG_RING, nullptr);
    *out = c;
    return 0;
}

int trhip_ipc_export(trhip_ipc* c, void* blob_out) {
    if (!c || !blob_out) return ipc_fail("trhip_ipc_export: null argument");
    ICHK(hipSetDevice(c->device));
    Blob b;
    memset(&b, 0, sizeof(b));
    memcpy(b.magic, "TRHIPIPC", 8);
    b.rank = c->rank; b.root = c->root; b.nranks = c->nranks; b.slots = c->slots; b.slot_bytes = c->slot_bytes; b.has_arena = c->arena != nullptr;
    if (c->arena) ICHK(hipIpcGetMemHandle(&b.arena, c->arena));
    ICHK(hipIpcGetMemHandle(&b.tags, c->tags));
    memset(blob_out, 0, TRHIP_IPC_EXPORT_BYTES);
    memcpy(blob_out, &b, sizeof(b));
    return 0;
}

int trhip_ipc_connect(trhip_ipc* c, const void* blobs) {
    if (!c || !blobs) return ipc_fail("trhip_ipc_connect: null argument");
    if (c->connected) return ipc_fail("trhip_ipc_connect: already connected");
    ICHK(hipSetDevice(c->device));
    const char* base = static_cast<const char*>(blobs);
    std::vector<Blob> all((size_t)c->nranks);
    for (int r = 0; r < c->nranks; ++r) {
        memcpy(&all[(size_t)r], base + (size_t)r * TRHIP_IPC_EXPORT_BYTES, sizeof(Blob));
        const Blob& b = all[(size_t)r];
        if (memcmp(b.magic, "TRHIPIPC", 8) != 0 || b.rank != r || b.root != c->root || b.nranks != c->nranks || b.slots != c->slots || b.slot_bytes != c->slot_bytes)
            return ipc_fail("trhip_ipc_connect: the blob of rank " + std::to_string(r) + " does not belong to this exchange (rank order, root, slots or slot size differ)");
    }
    if (c->rank == c->root) {
        c->peer_tags.assign((size_t)c->nranks, nullptr);
        for (int r = 0; r < c->nranks; ++r) {
            if (r == c->root) continue;
            void* p = nullptr;
            ICHK(hipIpcOpenMemHandle(&p, all[(size_t)r].tags, hipIpcMemLazyEnablePeerAccess));
            c->peer_tags[(size_t)r] = static_cast<unsigned long long*>(p);
        }
    } else {
        const Blob& rb = all[(size_t)c->root];
        if (!rb.has_arena) return ipc_fail("trhip_ipc_connect: the root exported no arena");
        ICHK(hipIpcOpenMemHandle(&c->root_arena, rb.arena, hipIpcMemLazyEnablePeerAccess));
        void* p = nullptr;
        ICHK(hipIpcOpenMemHandle(&p, rb.tags, hipIpcMemLazyEnablePeerAccess));
        c->root_tags = static_cast<unsigned long long*>(p);
    }
    c->connected = true;
    return 0;
}

int trhip_ipc_gather_partials(trhip_ipc* c, const void* send_dev, size_t send_bytes, void** recv_dev_out, const size_t* recv_bytes, void* stream) {
    if (!c) return ipc_fail("trhip_ipc_gather_partials: null exchange");
    if (c->nranks == 1) return 0;
    if (!c->connected) return ipc_fail("trhip_ipc_gather_partials: call trhip_ipc_connect first");
    if (int rc = trhip_ipc_check(c)) return rc;
    // every argument is checked before the frame counter moves: a call that fails leaves this rank's slot and tag sequence where the
    // other ranks expect them
    if (c->rank != c->root) {
        if (send_bytes > c->slot_bytes) return ipc_fail("trhip_ipc_gather_partials: the partial frame is larger than a slot");
        if (send_bytes != 0 && !send_dev) return ipc_fail("trhip_ipc_gather_partials: null send buffer");
    } else {
        if (!recv_dev_out || !recv_bytes) return ipc_fail("trhip_ipc_gather_partials: the root needs the receive arrays");
        for (int r = 0; r < c->nranks; ++r)
            if (r != c->root && recv_bytes[r] > c->slot_bytes) return ipc_fail("trhip_ipc_gather_partials: a partial frame is larger than a slot");
    }
    ICHK(hipSetDevice(c->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned long long f = c->frame;
    const size_t k = (size_t)(f % TAG_RING);
    if (int rc = ring_acquire(c, k)) return rc;
    c->frame++;
    const int slot = (int)(f % (unsigned long long)c->slots);
    const unsigned long long use = f / (unsigned long long)c->slots + 1;      // the slot's use count with this frame: its tag value
    unsigned long long* val = c->tag_values + k;
    unsigned char* wanted = c->wanted_ring + 64 * k;
    if (c->rank != c->root) {
        if (send_bytes == 0) return 0;      // (a rank whose share is empty sends nothing and is not waited for: recv_bytes[r] = 0 on the root)
        if (use > 1) {      // the slot's previous frame has to be consumed: release tag >= use - 1
            wanted[0] = 1;
            ICHK(hipMemcpyAsync(c->wanted_dev + 64 * slot, wanted, 1, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_wait_tags, dim3(1), dim3(64), 0, s, c->tags + slot, 1, 1, c->wanted_dev + 64 * slot, use - 1, c->timed_out, c->timeout_ticks);
            ICHK(hipGetLastError());
        }
        char* dst = static_cast<char*>(c->root_arena) + ((size_t)slot * c->nranks + (size_t)c->rank) * c->slot_bytes;
        ICHK(hipMemcpyAsync(dst, send_dev, send_bytes, hipMemcpyDeviceToDevice, s));
        *val = use;
        ICHK(hipMemcpyAsync(c->root_tags + (size_t)c->rank * c->slots + slot, val, 8, hipMemcpyHostToDevice, s));
        return ring_release(c, k, s);
    }
    bool any = false;
    for (int r = 0; r < c->nranks; ++r) {
        const bool w = r != c->root && recv_bytes[r] > 0;
        wanted[r] = w ? 1 : 0;
        recv_dev_out[r] = w ? static_cast<char*>(c->arena) + ((size_t)slot * c->nranks + (size_t)r) * c->slot_bytes : nullptr;
        any |= w;
    }
    if (any) {
        ICHK(hipMemcpyAsync(c->wanted_dev + 64 * slot, wanted, (size_t)c->nranks, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_wait_tags, dim3(1), dim3(64), 0, s, c->tags + slot, c->nranks, c->slots, c->wanted_dev + 64 * slot, use, c->timed_out, c->timeout_ticks);
        ICHK(hipGetLastError());
        return ring_release(c, k, s);
    }
    return 0;
}

// A device-side wait that gave up reports through pinned memory; this turns it into an error.  Every call into the exchange begins with it,
// and a caller about to use a frame (save it, show it) asks once more after synchronising that frame's stream: the last frame of a job
// has no next call.
int trhip_ipc_check(trhip_ipc* c) {
    if (!c) return ipc_fail("trhip_ipc_check: null exchange");
    if (c->timed_out && *static_cast<volatile int*>(c->timed_out))
        return ipc_fail("trhip_ipc: a frame's wait for a peer gave up (a rank died or fell " + std::to_string(c->timeout_ticks / 100000ull) + " ms behind): that frame is incomplete");
    return 0;
}

int trhip_ipc_release(trhip_ipc* c, void* stream) {
    if (!c) return ipc_fail("trhip_ipc_release: null exchange");
    if (c->nranks == 1 || c->rank != c->root) return 0;
    if (c->frame == 0) return ipc_fail("trhip_ipc_release: nothing gathered yet");
    if (int rc = trhip_ipc_check(c)) return rc;
    ICHK(hipSetDevice(c->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned long long f = c->frame - 1;
    const int slot = (int)(f % (unsigned long long)c->slots);
    const size_t k = TAG_RING + (size_t)(c->releases % TAG_RING);      // the releases have ring entries of their own
    if (int rc = ring_acquire(c, k)) return rc;
    c->releases++;
    unsigned long long* val = c->tag_values + k;
    *val = f / (unsigned long long)c->slots + 1;
    for (int r = 0; r < c->nranks; ++r)
        if (r != c->root && c->peer_tags[(size_t)r]) ICHK(hipMemcpyAsync(c->peer_tags[(size_t)r] + slot, val, 8, hipMemcpyHostToDevice, s));
    return ring_release(c, k, s);
}

void trhip_ipc_destroy(trhip_ipc* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->root_arena) (void)hipIpcCloseMemHandle(c->root_arena);
    if (c->root_tags) (void)hipIpcCloseMemHandle(c->root_tags);
    for (unsigned long long* p : c->peer_tags) if (p) (void)hipIpcCloseMemHandle(p);
    if (c->arena) (void)hipFree(c->arena);
    if (c->tags) (void)hipFree(c->tags);
    if (c->timed_out) (void)hipHostFree(c->timed_out);
    if (c->wanted_dev) (void)hipFree(c->wanted_dev);
    if (c->tag_values) (void)hipHostFree(c->tag_values);
    if (c->wanted_ring) (void)hipHostFree(c->wanted_ring);
    for (hipEvent_t e : c->ring_events) if (e) (void)hipEventDestroy(e);
    delete c;
}

}  // extern "C"
