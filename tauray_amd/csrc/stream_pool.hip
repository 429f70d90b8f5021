// Streams of a device, kept for the life of the process and handed out by the hardware pipe they landed on.
//
// What was measured (tools/debug/strip_order_probe.py, extra_streams_probe.py; profiles/r5/stream_pipes.txt): the compute queues of a
// process sit on four hardware pipes, the n-th queue created on pipe n mod 4, and two queues of one pipe do not overlap launches that
// do not fit the chip at once - the persistent grids of this path tracer: the second launch waits until the first one's last
// workgroup has been placed.  Four lanes on four pipes render a 1/8 strip of sponza_teapots in 0.75 ms and the whole frame in 3.7;
// with two of the lanes on one pipe it is 1.0 and 4.0-4.5 ms - and which pipes a renderer's streams get depends on every stream the
// process created before (other renderers, the application's own, torch's, RCCL's).  So streams are never destroyed here, every
// stream is classified once by an experiment (a launch that holds its pipe for ~0.3 ms, a one-wave launch on the other stream: does
// it start at once?), and a taker says which streams its new stream must be able to overlap with.
//
// The contract (round 6; include/trhip.h "process requirements"):
//   * class representatives are streams this file created - never a caller's handle, which may be destroyed behind our back;
//   * a caller's (foreign) stream is classified against them at most once per stream *identity* (handle + hipStreamGetId; the null
//     stream is its own identity), and only when that is harmless: stream_pool_class(s, .., blocking = true) - the explicit
//     trhip_stream_pipe_class - synchronises `s`; every other path (a stage meeting a new caller's stream in render()) probes only a
//     stream that is idle and not capturing, and otherwise reports class -1 (unknown) without remembering it;
//   * every HIP call of the experiment is checked; a failed or timed-out experiment yields "unknown", never a new class;
//   * "slow" is relative to the duration of the pipe-holding launch measured in the same trial, not a number of microseconds;
//   * there are at most MAX_CLASSES classes; whatever does not fit is class -1;
//   * TRHIP_PIPE_PROBE=0: no experiment, every stream class -1.  TRHIP_PIPE_CLASSES=0,1,2,3: no experiment either; the n-th stream
//     created here is of class list[n mod len] (what the experiment finds on gfx950 + ROCm 7.2 with GPU_MAX_HW_QUEUES=8 is 0,1,2,3),
//     foreign streams are class -1.
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "pt.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return set_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

namespace tr {
namespace {

constexpr int MAX_CLASSES = 8;

__global__ __launch_bounds__(64) void k_pipe_hog(int* started, unsigned ticks) {
    __shared__ int pad[16000];       // 64 000 bytes: two blocks per CU, so a grid of many rounds stays in its pipe while it runs
    pad[threadIdx.x] = (int)blockIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) { __hip_atomic_store(started, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    const unsigned long long t0 = wall_clock64();       // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (pad[63 - threadIdx.x] == -7) __hip_atomic_store(started, 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // keeps the array
}
__global__ void k_pipe_mark(int* out) { if (threadIdx.x == 0) *out = 1; }

struct Pooled { hipStream_t s; int cls; bool free; };
struct Foreign { hipStream_t s; unsigned long long id; int cls; };
struct DevicePool {
    bool init = false;
    bool probe = true;
    std::vector<int> pinned;    // TRHIP_PIPE_CLASSES
    int* flag = nullptr;        // host memory the device writes
    int* mark = nullptr;        // device word
    int cus = 256;
    int classes = 0;
    bool debug = false;
    bool warned_overflow = false;
    std::vector<Pooled> streams;                 // created here
    std::vector<Foreign> foreign;                // the caller's streams (the null stream among them), by identity
    size_t reps[MAX_CLASSES] = {};               // one stream of `streams` per class
};
std::mutex g_mutex;
DevicePool g_pools[16];

double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

enum { PIPE_DIFFERENT = 0, PIPE_SAME = 1, PIPE_UNKNOWN = 2 };

// PIPE_SAME when a launch on `b` cannot start while a launch on `a` is still placing its workgroups.  `a` is a stream of the pool
// (synchronised here without asking: it is ours); `b` is idle by the caller's word or is synchronised because the caller asked for it.
int same_pipe(DevicePool& P, hipStream_t a, hipStream_t b) {
    if (a == b) return PIPE_SAME;
    // both idle first: work already queued on either would be taken for the other launch holding the pipe
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) { (void)hipGetLastError(); return PIPE_UNKNOWN; }
    int slow = 0;
    for (int trial = 0; trial < 2; ++trial) {
        *(volatile int*)P.flag = 0;
        hipLaunchKernelGGL(k_pipe_hog, dim3((unsigned)P.cus * 2u * 12u), dim3(64), 0, a, P.flag, 2500u);      // twelve rounds of 25 us
        if (hipGetLastError() != hipSuccess) return PIPE_UNKNOWN;
        const double t_wait = now_us();
        while (*(volatile int*)P.flag == 0 && now_us() - t_wait < 50000.0) {}
        if (*(volatile int*)P.flag == 0) {      // the launch never started (a wedged queue, a device under another process's load)
            if (hipStreamSynchronize(a) != hipSuccess) (void)hipGetLastError();
            return PIPE_UNKNOWN;
        }
        const double t0 = now_us();
        hipLaunchKernelGGL(k_pipe_mark, dim3(1), dim3(64), 0, b, P.mark);
        if (hipGetLastError() != hipSuccess) { (void)hipStreamSynchronize(a); (void)hipGetLastError(); return PIPE_UNKNOWN; }
        if (hipStreamSynchronize(b) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(a); (void)hipGetLastError(); return PIPE_UNKNOWN; }
        const double dt = now_us() - t0;
        if (hipStreamSynchronize(a) != hipSuccess) { (void)hipGetLastError(); return PIPE_UNKNOWN; }
        const double hog = now_us() - t0;      // what was left of the pipe-holding launch when the mark was enqueued (~0.3 ms)
        // one pipe: the mark starts when the hog's last round is placed, ~11/12 of `hog`; two pipes: it starts at once (10-20 us)
        if (dt > 0.4 * hog && dt > 40.0) slow++;
        else return PIPE_DIFFERENT;              // one prompt start settles it: the pipes differ
    }
    return slow == 2 ? PIPE_SAME : PIPE_UNKNOWN;
}

// the class of stream `s` by experiment against the representatives; `own` >= 0: s is streams[own] and may found a new class
int classify(DevicePool& P, hipStream_t s, int own) {
    if (!P.probe) return -1;
    for (int c = 0; c < P.classes; ++c) {
        const int r = same_pipe(P, P.streams[P.reps[c]].s, s);
        if (r == PIPE_SAME) return c;
        if (r == PIPE_UNKNOWN) return -1;
    }
    if (own < 0) return -2;      // a foreign stream on a pipe none of ours sits on (it cannot be a representative itself)
    if (P.classes >= MAX_CLASSES) {
        if (P.debug && !P.warned_overflow) { fprintf(stderr, "trhip: more than %d pipe classes found; further streams are unclassified\n", MAX_CLASSES); P.warned_overflow = true; }
        return -1;
    }
    P.reps[P.classes] = (size_t)own;
    return P.classes++;
}

int ensure_init(DevicePool& P, int dev) {
    if (P.init) return 0;
    P.probe = !(getenv("TRHIP_PIPE_PROBE") && atoi(getenv("TRHIP_PIPE_PROBE")) == 0);
    P.debug = getenv("TRHIP_DEBUG") && atoi(getenv("TRHIP_DEBUG")) != 0;
    if (const char* pc = getenv("TRHIP_PIPE_CLASSES")) {
        for (const char* p = pc; *p;) {
            char* end = nullptr;
            const long v = strtol(p, &end, 10);
            if (end == p) break;
            if (v >= 0 && v < MAX_CLASSES) P.pinned.push_back((int)v);
            p = *end ? end + 1 : end;
        }
        if (!P.pinned.empty()) { P.probe = false; for (int v : P.pinned) P.classes = std::max(P.classes, v + 1); }
    }
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&P.flag), 64, hipHostMallocDefault));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&P.mark), 64));
    (void)hipDeviceGetAttribute(&P.cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (P.cus <= 0) P.cus = 256;
    P.init = true;
    return 0;
}

int make_stream(DevicePool& P, bool free_after, hipStream_t* out) {
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int idx = (int)P.streams.size();
    P.streams.push_back(Pooled{s, -1, free_after});
    P.streams[idx].cls = !P.pinned.empty() ? P.pinned[(size_t)idx % P.pinned.size()] : classify(P, s, idx);
    *out = s;
    return 0;
}

// hipStreamGetId exists from ROCm 7.1; a process that mapped an older libamdhip64 first (torch 2.10 brings its own 7.0) must still be able
// to load this library, so the symbol is looked up at run time.  Without it a foreign stream has no identity beyond its handle, which
// a new stream can inherit: such streams are then classified every time they are asked about and never remembered (NO_IDENTITY).
constexpr unsigned long long NOT_A_STREAM = ~0ull, NO_IDENTITY = ~0ull - 1;
unsigned long long identity_of(hipStream_t s) {
    if (!s) return 0;
    typedef hipError_t (*get_id_fn)(hipStream_t, unsigned long long*);
    static const get_id_fn get_id = reinterpret_cast<get_id_fn>(dlsym(RTLD_DEFAULT, "hipStreamGetId"));
    if (!get_id) return hipStreamQuery(s) == hipErrorInvalidHandle ? ((void)hipGetLastError(), NOT_A_STREAM) : ((void)hipGetLastError(), NO_IDENTITY);
    unsigned long long id = 0;
    if (get_id(s, &id) != hipSuccess) { (void)hipGetLastError(); return NOT_A_STREAM; }
    return id;
}

// blocking: the caller allows `s` to be synchronised.  Otherwise a stream that is busy or capturing is not touched.
int class_of(DevicePool& P, hipStream_t s, bool blocking) {
    for (auto& e : P.streams) if (e.s == s) return e.cls;
    if (!P.probe) return -1;
    const unsigned long long id = identity_of(s);
    if (id == NOT_A_STREAM) return -1;      // not a live stream of this process
    if (id != NO_IDENTITY) for (auto& e : P.foreign) if (e.s == s && e.id == id) return e.cls;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (s && (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) { (void)hipGetLastError(); return -1; }
    if (!blocking && hipStreamQuery(s) != hipSuccess) { (void)hipGetLastError(); return -1; }      // busy: not ours to wait for
    if (P.streams.empty()) { hipStream_t first = nullptr; if (make_stream(P, true, &first)) return -1; }      // something to compare with
    int c = classify(P, s, -1);
    // on a pipe none of our streams sits on yet: make streams (kept, idle) until one lands there - eight cover every hardware queue once
    while (c == -2 && P.streams.size() < 8) {
        hipStream_t more = nullptr;
        if (make_stream(P, true, &more)) return -1;
        c = classify(P, s, -1);
    }
    if (c < 0) c = -1;
    if (c >= 0 && id != NO_IDENTITY) {
        for (auto it = P.foreign.begin(); it != P.foreign.end();) it = it->s == s ? P.foreign.erase(it) : it + 1;      // the handle has a new owner
        P.foreign.push_back(Foreign{s, id, c});
    }
    return c;
}

}  // namespace

int stream_pool_acquire(hipStream_t* out, const hipStream_t* overlap_with, int n_overlap) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mutex);
    DevicePool& P = g_pools[dev & 15];
    if (int rc = ensure_init(P, dev)) return rc;
    unsigned avoid = 0;
    for (int i = 0; i < n_overlap; ++i) { const int c = class_of(P, overlap_with[i], false); if (c >= 0) avoid |= 1u << c; }
    int in_use[MAX_CLASSES] = {};
    for (auto& e : P.streams) if (!e.free && e.cls >= 0) in_use[e.cls]++;
    for (int round = 0; round < 2; ++round) {
        // the idle stream whose pipe has the fewest takers, the oldest among equals
        int best = -1;
        for (size_t i = 0; i < P.streams.size(); ++i) {
            const Pooled& e = P.streams[i];
            if (!e.free || (e.cls >= 0 && (avoid >> e.cls & 1u))) continue;
            if (best < 0 || (e.cls >= 0 && P.streams[best].cls >= 0 && in_use[e.cls] < in_use[P.streams[best].cls])) best = (int)i;
        }
        if (best >= 0) { P.streams[best].free = false; *out = P.streams[best].s; return 0; }
        if (round == 1) break;
        // none: make streams until one lands on a pipe that is not taken (eight streams cover four pipes twice over)
        for (int made = 0; made < 8 && P.streams.size() < 24; ++made) {
            hipStream_t s = nullptr;
            if (int rc = make_stream(P, true, &s)) return rc;
            const int c = P.streams.back().cls;
            if (c < 0 || !(avoid >> c & 1u)) break;
        }
    }
    // every pipe is taken by the streams named: any idle stream, or one more
    for (auto& e : P.streams) if (e.free) { e.free = false; *out = e.s; return 0; }
    return make_stream(P, false, out);
}

void stream_pool_release(hipStream_t s) {
    if (!s) return;
    (void)hipStreamSynchronize(s);
    std::lock_guard<std::mutex> lock(g_mutex);
    for (auto& P : g_pools) for (auto& e : P.streams) if (e.s == s) { e.free = true; return; }
    (void)hipStreamDestroy(s);       // not one of ours
}

int stream_pool_class(hipStream_t s, int* cls, bool blocking) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mutex);
    DevicePool& P = g_pools[dev & 15];
    if (int rc = ensure_init(P, dev)) return rc;
    *cls = class_of(P, s, blocking);
    return 0;
}

// How many pipe classes this process can reach on the current device: streams are made (and kept, idle) until eight exist, which
// covers the hardware queues of GPU_MAX_HW_QUEUES = 8 once.  0 with the experiment switched off.
int stream_pool_pipe_classes(int* classes_out, int* streams_out) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mutex);
    DevicePool& P = g_pools[dev & 15];
    if (int rc = ensure_init(P, dev)) return rc;
    if (P.probe) while (P.streams.size() < 8) { hipStream_t s = nullptr; if (int rc = make_stream(P, true, &s)) return rc; }
    *classes_out = P.classes;
    if (streams_out) *streams_out = (int)P.streams.size();
    return 0;
}

}  // namespace tr
