// Streams of a device, kept for the life of the process and handed out by the hardware pipe they landed on.
//
// What was measured (tools/debug/strip_order_probe.py, extra_streams_probe.py; profiles/r5/stream_pipes.txt): the compute queues of a
// process sit on four hardware pipes, the n-th queue created on pipe n mod 4, and two queues of one pipe do not overlap launches that
// do not fit the chip at once - the persistent grids of this path tracer: the second launch waits until the first one's last
// workgroup has been placed.  Four lanes on four pipes render a 1/8 strip of sponza_teapots in 0.75 ms and the whole frame in 3.7;
// with two of the lanes on one pipe it is 1.0 and 4.0-4.5 ms - and which pipes a renderer's streams get depends on every stream the
// process created before (other renderers, the application's own, torch's, RCCL's).  So streams are never destroyed here, every
// stream is classified once by an experiment (a launch that holds its pipe for ~0.3 ms, a one-wave launch on the other stream: does
// it start at once?), and a taker says which streams its new stream must be able to overlap with.  TRHIP_PIPE_PROBE=0 turns the
// experiment off (every stream its own class).
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
struct DevicePool {
    bool init = false;
    bool probe = true;
    int* flag = nullptr;        // host memory the device writes
    int* mark = nullptr;        // device word
    int cus = 256;
    std::vector<Pooled> streams;                 // created here
    std::vector<std::pair<hipStream_t, int>> foreign;   // streams of the caller (the null stream among them) seen as a main stream
    std::vector<hipStream_t> reps;               // one stream per class
};
std::mutex g_mutex;
DevicePool g_pools[16];

double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// true when a launch on `b` cannot start while a launch on `a` is still placing its workgroups
bool same_pipe(DevicePool& P, hipStream_t a, hipStream_t b) {
    if (a == b) return true;
    // both idle first: work already queued on either (a caller's stream seen for the first time in the middle of its frame) would be
    // taken for the other launch holding the pipe
    (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
    int slow = 0;
    for (int trial = 0; trial < 2; ++trial) {
        *(volatile int*)P.flag = 0;
        hipLaunchKernelGGL(k_pipe_hog, dim3((unsigned)P.cus * 2u * 12u), dim3(64), 0, a, P.flag, 2500u);      // twelve rounds of 25 us
        const double t_wait = now_us();
        while (*(volatile int*)P.flag == 0 && now_us() - t_wait < 50000.0) {}
        const double t0 = now_us();
        hipLaunchKernelGGL(k_pipe_mark, dim3(1), dim3(64), 0, b, P.mark);
        (void)hipStreamSynchronize(b);
        const double dt = now_us() - t0;
        (void)hipStreamSynchronize(a);
        if (dt > 120.0) slow++;
        else break;              // one prompt start settles it: the pipes differ
    }
    return slow == 2;
}

int classify(DevicePool& P, hipStream_t s) {
    if (!P.probe) return -1;
    for (size_t c = 0; c < P.reps.size(); ++c) if (same_pipe(P, P.reps[c], s)) return (int)c;
    P.reps.push_back(s);
    return (int)P.reps.size() - 1;
}

int ensure_init(DevicePool& P, int dev) {
    if (P.init) return 0;
    P.probe = !(getenv("TRHIP_PIPE_PROBE") && atoi(getenv("TRHIP_PIPE_PROBE")) == 0);
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&P.flag), 64, hipHostMallocDefault));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&P.mark), 64));
    (void)hipDeviceGetAttribute(&P.cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (P.cus <= 0) P.cus = 256;
    P.init = true;
    return 0;
}

int class_of(DevicePool& P, hipStream_t s) {
    for (auto& e : P.streams) if (e.s == s) return e.cls;
    for (auto& e : P.foreign) if (e.first == s) return e.second;
    const int c = classify(P, s);
    P.foreign.emplace_back(s, c);
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
    for (int i = 0; i < n_overlap; ++i) { const int c = class_of(P, overlap_with[i]); if (c >= 0) avoid |= 1u << c; }
    int in_use[32] = {};
    for (auto& e : P.streams) if (!e.free && e.cls >= 0) in_use[e.cls & 31]++;
    for (int round = 0; round < 2; ++round) {
        // the idle stream whose pipe has the fewest takers, the oldest among equals
        int best = -1;
        for (size_t i = 0; i < P.streams.size(); ++i) {
            const Pooled& e = P.streams[i];
            if (!e.free || (e.cls >= 0 && (avoid >> e.cls & 1u))) continue;
            if (best < 0 || (e.cls >= 0 && P.streams[best].cls >= 0 && in_use[e.cls & 31] < in_use[P.streams[best].cls & 31])) best = (int)i;
        }
        if (best >= 0) { P.streams[best].free = false; *out = P.streams[best].s; return 0; }
        if (round == 1) break;
        // none: make streams until one lands on a pipe that is not taken (eight streams cover four pipes twice over)
        for (int made = 0; made < 8 && P.streams.size() < 24; ++made) {
            hipStream_t s = nullptr;
            HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            const int c = classify(P, s);
            P.streams.push_back(Pooled{s, c, true});
            if (c < 0 || !(avoid >> c & 1u)) break;
        }
    }
    // every pipe is taken by the streams named: any idle stream, or one more
    for (auto& e : P.streams) if (e.free) { e.free = false; *out = e.s; return 0; }
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    P.streams.push_back(Pooled{s, P.probe ? classify(P, s) : -1, false});
    *out = s;
    return 0;
}

void stream_pool_release(hipStream_t s) {
    if (!s) return;
    (void)hipStreamSynchronize(s);
    std::lock_guard<std::mutex> lock(g_mutex);
    for (auto& P : g_pools) for (auto& e : P.streams) if (e.s == s) { e.free = true; return; }
    (void)hipStreamDestroy(s);       // not one of ours
}

int stream_pool_class(hipStream_t s, int* cls) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mutex);
    DevicePool& P = g_pools[dev & 15];
    if (int rc = ensure_init(P, dev)) return rc;
    *cls = class_of(P, s);
    return 0;
}

}  // namespace tr
