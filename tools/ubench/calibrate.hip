// Calibration kernels for the roofline of bench.py (gfx950).  Every kernel does a known amount of work - wave-level vector
// instructions, bytes requested from L2, bytes that must come from beyond L2 - so that the rocprofv3 counters the bench
// reports (SQ_INSTS_VALU, TCP_TCC_READ_REQ_sum, TCC_EA0_RDREQ_sum / FETCH_SIZE, ...) can be turned into instructions and bytes
// with measured factors instead of assumed ones, and so that the peaks the fractions are taken against are measured on the box.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/calibrate tools/ubench/calibrate.hip
//   tools/ubench/calibrate [reps]                               -> one JSON line per kernel (known work, HIP-event time)
//   rocprofv3 --pmc <counters> --kernel-trace ... -- calibrate  -> per-dispatch counters of the same kernels (tools/calibration_summary.py)
//
// Access patterns are the ones the trace kernels use: a node visit is seven dwordx4 loads of one lane from one 128-byte line
// (k_cal_gather_node), rays / hits / path state are 16-byte-per-lane coalesced streams (k_cal_stream16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- VALU issue: ITER x 16 independent v_fma_f32 per wave, nothing else in the loop but the counter
template <int UNROLL>
__global__ __launch_bounds__(256) void k_cal_fma(float* out, int iters, float a, float b) {
    float r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = (float)threadIdx.x + (float)k;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[k]) : "v"(a), "v"(b));
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += r[k];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// the instruction mix of a slab test: sub, mul, max, min, compare + select (all full-rate fp32 / integer VALU forms)
__global__ __launch_bounds__(256) void k_cal_mix(float* out, int iters, float a, float b) {
    float r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = (float)threadIdx.x + (float)k;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
            asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[k]) : "v"(a));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[k + 1]) : "v"(b));
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[k + 2]) : "v"(a));
            asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[k + 3]) : "v"(b));
        }
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
            asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[k]) : "v"(r[k + 1]), "v"(r[k + 2]) : "vcc");
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[k + 1]) : "v"(a), "v"(b));
            asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[k + 2]) : "v"(r[k + 3]));
            asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[k + 3]));
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += r[k];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// ---- coalesced 16-byte-per-lane stream: every lane reads `n16` float4 with a grid stride; `passes` sweeps over the buffer
__global__ __launch_bounds__(256) void k_cal_stream16(const float4* __restrict__ in, size_t n16, int passes, float* out) {
    float s = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(in) + i);
            s += v.x + v.y + v.z + v.w;
        }
    if (s == 12345.678f) out[threadIdx.x] = s;
}
// the same with ordinary (cached) loads
__global__ __launch_bounds__(256) void k_cal_stream16c(const float4* __restrict__ in, size_t n16, int passes, float* out) {
    float s = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
            const float4 v = in[i];
            s += v.x + v.y + v.z + v.w;
        }
    if (s == 12345.678f) out[threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_cal_stream4(const float* __restrict__ in, size_t n4, int passes, float* out) {
    float s = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) s += in[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_cal_write16(float4* __restrict__ o, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) o[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}

// ---- a node visit: the lane picks a pseudo-random 128-byte line of the buffer and reads seven dwordx4 from it (112 bytes);
// `visits` dependent visits per lane (the next line depends on the data just read, like a traversal)
__device__ __forceinline__ uint32_t pcg(uint32_t v) { uint32_t s = v * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u; return (w >> 22u) ^ w; }
__global__ __launch_bounds__(256) void k_cal_gather_node(const char* __restrict__ base, uint32_t lines, int visits, float* out) {
    uint32_t h = pcg(blockIdx.x * 256u + threadIdx.x + 1u);
    float s = 0;
    for (int v = 0; v < visits; ++v) {
        const uint32_t line = h % lines;
        const float4* p = reinterpret_cast<const float4*>(base + ((size_t)line << 7));
        const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6];
        s += a.x + b.y + c.z + d.w + e.x + f.y + g.z;
        h = pcg(h + (uint32_t)__float_as_uint(g.w));     // the buffer holds zeros: the chain stays the host's pcg chain
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// The same gather with NLOAD dwordx4 per visit from nodes of NODE_BYTES (what would a smaller node buy: 64-byte nodes read with
// four loads, 32-byte ones with two?).  The chain is dependent like a traversal.
template <int NLOAD, int NODE_SHIFT>
__global__ __launch_bounds__(256) void k_cal_gather_var(const char* __restrict__ base, uint32_t nodes, int visits, float* out) {
    uint32_t h = pcg(blockIdx.x * 256u + threadIdx.x + 1u);
    float s = 0;
    for (int v = 0; v < visits; ++v) {
        const uint32_t node = h % nodes;
        const float4* p = reinterpret_cast<const float4*>(base + ((size_t)node << NODE_SHIFT));
        float4 last = p[0];
        s += last.x;
#pragma unroll
        for (int k = 1; k < NLOAD; ++k) { const float4 q = p[k]; s += q.y; last = q; }
        h = pcg(h + (uint32_t)__float_as_uint(last.w));
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <class F>
static double time_ms(int reps, F&& launch) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    launch();                       // warm-up
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) launch();
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, a, b));
    CHK(hipEventDestroy(a)); CHK(hipEventDestroy(b));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clock_ghz = prop.clockRate / 1e6;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f}\n", prop.gcnArchName, cus, clock_ghz);
    float* out;
    CHK(hipMalloc(&out, 4096));

    // ---- VALU: 8 blocks of 256 threads per CU = 8 waves per SIMD (and 1, 2, 4 waves per SIMD for the latency-bound ends)
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * wps, iters = 4096;
        const double ms = time_ms(reps, [&] { hipLaunchKernelGGL(k_cal_fma<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); });
        const double winst = (double)blocks * 4.0 * iters * 64.0;       // wave-level v_fma_f32 instructions
        printf("{\"kernel\": \"k_cal_fma\", \"waves_per_simd\": %d, \"wave_insts\": %.0f, \"ms\": %.4f, \"ginst_per_s\": %.1f, \"cycles_per_inst_at_nominal_clock\": %.3f}\n",
               wps, winst, ms, winst / ms / 1e6, (double)cus * 4.0 * clock_ghz * 1e9 * ms * 1e-3 / winst);
    }
    {
        const int blocks = cus * 8, iters = 8192;
        const double ms = time_ms(reps, [&] { hipLaunchKernelGGL(k_cal_mix, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); });
        const double winst = (double)blocks * 4.0 * iters * 36.0;       // 16 + 4 * 5 vector instructions per iteration
        printf("{\"kernel\": \"k_cal_mix\", \"waves_per_simd\": 8, \"wave_insts\": %.0f, \"ms\": %.4f, \"ginst_per_s\": %.1f, \"cycles_per_inst_at_nominal_clock\": %.3f}\n",
               winst, ms, winst / ms / 1e6, (double)cus * 4.0 * clock_ghz * 1e9 * ms * 1e-3 / winst);
    }

    // ---- streams: 2 GiB (beyond the 256 MiB Infinity Cache), 128 MiB (inside it, beyond the 32 MiB of L2), 2 MiB (inside every L2)
    const size_t big = (size_t)2 << 30;
    char* buf;
    CHK(hipMalloc(&buf, big));
    CHK(hipMemset(buf, 0, big));
    struct S { const char* tag; size_t bytes; int passes; };
    for (const S& s : {S{"hbm_2GiB", big, 1}, S{"mall_128MiB", (size_t)128 << 20, 16}, S{"l2_2MiB", (size_t)2 << 20, 1024}}) {
        const size_t n16 = s.bytes / 16;
        const int blocks = cus * 8;
        double ms = time_ms(reps, [&] { hipLaunchKernelGGL(k_cal_stream16, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const float4*>(buf), n16, s.passes, out); });
        printf("{\"kernel\": \"k_cal_stream16\", \"footprint\": \"%s\", \"bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.1f}\n", s.tag, (double)s.bytes * s.passes, ms, (double)s.bytes * s.passes / ms / 1e6);
        ms = time_ms(reps, [&] { hipLaunchKernelGGL(k_cal_stream16c, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const float4*>(buf), n16, s.passes, out); });
        printf("{\"kernel\": \"k_cal_stream16c\", \"footprint\": \"%s\", \"bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.1f}\n", s.tag, (double)s.bytes * s.passes, ms, (double)s.bytes * s.passes / ms / 1e6);
        ms = time_ms(reps, [&] { hipLaunchKernelGGL(k_cal_stream4, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const float*>(buf), s.bytes / 4, s.passes > 4 ? s.passes / 4 : 1, out); });
        printf("{\"kernel\": \"k_cal_stream4\", \"footprint\": \"%s\", \"bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.1f}\n", s.tag, (double)s.bytes * (s.passes > 4 ? s.passes / 4 : 1), ms,
               (double)s.bytes * (s.passes > 4 ? s.passes / 4 : 1) / ms / 1e6);
    }
    {
        const double ms = time_ms(reps, [&] { hipLaunchKernelGGL(k_cal_write16, dim3(cus * 8), dim3(256), 0, 0, reinterpret_cast<float4*>(buf), big / 16); });
        printf("{\"kernel\": \"k_cal_write16\", \"footprint\": \"hbm_2GiB\", \"bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.1f}\n", (double)big, ms, (double)big / ms / 1e6);
        CHK(hipMemset(buf, 0, big));
    }
    // ---- node gathers: 176 MB structure (sponza_teapots: 48 MB triangles + 128 MB nodes), 2 GiB, 2 MiB; six waves per SIMD like the closest-hit kernel
    for (const S& s : {S{"hbm_2GiB", big, 64}, S{"mall_128MiB", (size_t)128 << 20, 64}, S{"l2_2MiB", (size_t)2 << 20, 256}}) {
        const uint32_t lines = (uint32_t)(s.bytes >> 7);
        const int blocks = cus * 6;
        const double ms = time_ms(reps, [&] { hipLaunchKernelGGL(k_cal_gather_node, dim3(blocks), dim3(256), 0, 0, buf, lines, s.passes, out); });
        const double visits = (double)blocks * 256.0 * s.passes;
        printf("{\"kernel\": \"k_cal_gather_node\", \"footprint\": \"%s\", \"visits\": %.0f, \"bytes_112\": %.0f, \"bytes_lines_128\": %.0f, \"ms\": %.4f, \"Gvisits_per_s\": %.2f, \"GBps_112\": %.1f}\n",
               s.tag, visits, visits * 112.0, visits * 128.0, ms, visits / ms / 1e6, visits * 112.0 / ms / 1e6);
    }
    // ---- what the node size and the loads per visit cost: the structure of sponza_teapots holds 378 k live nodes of 128 bytes + 48 MB
    // of triangles; here the same NUMBER of nodes (1 M, 4 M) at 128 / 64 / 32 bytes each, six waves per SIMD
    {
        const int blocks = cus * 6, visits = 64;
        const double nv = (double)blocks * 256.0 * visits;
        for (uint32_t nodes : {1u << 20, 1u << 22}) {
            auto run = [&](const char* tag, int bytes, int loads, auto kernel) {
                const double ms = time_ms(reps, [&] { hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, buf, nodes, visits, out); });
                printf("{\"kernel\": \"k_cal_gather_var\", \"variant\": \"%s\", \"nodes\": %u, \"node_bytes\": %d, \"loads_per_visit\": %d, \"footprint_MB\": %.0f, \"visits\": %.0f, \"ms\": %.4f, \"Gvisits_per_s\": %.2f}\n",
                       tag, nodes, bytes, loads, (double)nodes * bytes / 1e6, nv, ms, nv / ms / 1e6);
            };
            run("128B_7loads", 128, 7, k_cal_gather_var<7, 7>);
            run("128B_4loads", 128, 4, k_cal_gather_var<4, 7>);
            run("128B_1load", 128, 1, k_cal_gather_var<1, 7>);
            run("64B_4loads", 64, 4, k_cal_gather_var<4, 6>);
            run("64B_2loads", 64, 2, k_cal_gather_var<2, 6>);
            run("32B_2loads", 32, 2, k_cal_gather_var<2, 5>);
        }
    }
    return 0;
}
