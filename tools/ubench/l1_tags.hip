// tools/ubench/l1_tags.hip - how many cache-line (tag) accesses per clock does the vector L1 (TCP) of a CU take?
// Independent `global_load_dwordx4`s out of an L1-resident footprint (16 KB, the same addresses for every wave), 1 / 4 / 8 / 64 lanes of
// a wave sharing a 128-byte line: 64 / 16 / 8 / 1 line accesses per wave instruction.  Known work + HIP-event time; run under
// `rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace` the counters say what
// they count per instruction.  The trace kernels' L1 level in bench.py's roofline is priced against the rate measured here
// (profiles/r3/l1_tag_rate.json).
//   hipcc --offload-arch=gfx950 -O3 -o l1_tags l1_tags.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// lane l reads 16 bytes at base + l * lane_stride (+ 8 KB for every other load): lane_stride 128 -> a line per lane, 32 -> four
// lanes per line, 16 -> eight (the contiguous 1 KB of a coalesced dwordx4), 0 -> one line for the wave.  UNROLL independent loads
// are in flight per wave and iteration.
template <int UNROLL>
__global__ __launch_bounds__(256) void k_cal_l1(const char* __restrict__ base, int iters, int lane_stride, float* out) {
    const int lane = threadIdx.x & 63;
    const char* p = base + (size_t)lane * (size_t)lane_stride;
    f4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const char* q = p + (u & 1) * 8192 + ((u >> 1) & 1) * 16 * (lane_stride == 128);   // two 8 KB halves; within a lane's line another 16 bytes
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[u]) : "v"(q) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    if (acc.x == 123.456f) out[0] = acc.y + acc.z + acc.w;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clock_ghz = prop.clockRate / 1e6;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f}\n", prop.gcnArchName, cus, clock_ghz);
    char* buf; float* out;
    CHK(hipMalloc(&buf, 1 << 20));
    CHK(hipMemset(buf, 0, 1 << 20));
    CHK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int strides[4] = {128, 32, 16, 0};
    const int lines[4] = {64, 16, 8, 1};
    for (int waves_per_simd : {2, 4, 8}) {
        for (int s = 0; s < 4; ++s) {
            const int blocks = cus * waves_per_simd;      // 256 threads = 4 waves = one per SIMD
            const int iters = s == 0 ? 400 : 2000;
            constexpr int UNROLL = 8;
            double best = 1e30;
            for (int r = 0; r < reps + 1; ++r) {
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_cal_l1<UNROLL>, dim3(blocks), dim3(256), 0, 0, buf, iters, strides[s], out);
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (r && ms < best) best = ms;
            }
            const double insts = (double)blocks * 4 * iters * UNROLL;
            const double acc = insts * lines[s];
            printf("{\"kernel\": \"k_cal_l1\", \"waves_per_simd\": %d, \"lanes_per_line\": %d, \"line_accesses_per_inst\": %d, \"wave_insts\": %.0f, \"line_accesses\": %.0f, \"ms\": %.4f, "
                   "\"Ginst_per_s\": %.2f, \"Gaccesses_per_s\": %.1f, \"accesses_per_clock_per_cu\": %.3f, \"insts_per_clock_per_cu\": %.4f, \"bytes_per_clock_per_cu\": %.1f}\n",
                   waves_per_simd, 64 / lines[s], lines[s], insts, acc, best, insts / best / 1e6, acc / best / 1e6, acc / (best * 1e-3) / cus / (clock_ghz * 1e9),
                   insts / (best * 1e-3) / cus / (clock_ghz * 1e9), insts * 1024.0 / (best * 1e-3) / cus / (clock_ghz * 1e9));
        }
    }
    return 0;
}
