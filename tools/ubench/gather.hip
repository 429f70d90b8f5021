// Micro-benchmark: cost of per-lane 64-byte record gathers on gfx950.
//  A: every lane issues 4 x dwordx4 for its own record (what the BVH2 traversal does)
//  B: quad-cooperative: in instruction j the 4 lanes of a quad read the 4 x 16 B of the record wanted by quad lane j
//     (one full 64-B line per quad per instruction), data returned to its owner with DPP-style shuffles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256) void k_a(const uint4* __restrict__ tab, const unsigned* __restrict__ idx, unsigned n_iter, unsigned mask, unsigned* out) {
    unsigned i = idx[blockIdx.x * 256 + threadIdx.x];
    unsigned acc = 0;
    for (unsigned it = 0; it < n_iter; ++it) {
        const uint4* p = tab + (size_t)(i & mask) * 4;
        uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x ^ b.y ^ c.z ^ d.w;
        i = i * 1664525u + 1013904223u + (a.x & 1u);   // dependent chain like a traversal
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_b(const uint4* __restrict__ tab, const unsigned* __restrict__ idx, unsigned n_iter, unsigned mask, unsigned* out) {
    unsigned i = idx[blockIdx.x * 256 + threadIdx.x];
    unsigned acc = 0;
    const int lane = threadIdx.x & 63, q = lane & 3, qbase = lane & ~3;
    for (unsigned it = 0; it < n_iter; ++it) {
        unsigned want = i & mask;
        uint4 got[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned rec = __shfl(want, qbase + j);          // record wanted by quad lane j
            got[j] = tab[(size_t)rec * 4 + q];                // my 16-B slice of it: the quad reads one 64-B line
        }
        // transpose back: I need slices 0..3 of MY record, slice c lives in lane qbase+c, in its got[q]
        uint4 mine[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint4 v;
                v.x = __shfl(got[j].x, qbase + c); v.y = __shfl(got[j].y, qbase + c); v.z = __shfl(got[j].z, qbase + c); v.w = __shfl(got[j].w, qbase + c);
                if (j == q) mine[c] = v;
            }
        }
        acc += mine[0].x ^ mine[1].y ^ mine[2].z ^ mine[3].w;
        i = i * 1664525u + 1013904223u + (mine[0].x & 1u);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// C: same memory pattern as B without the return transpose (upper bound of what the memory side can give)
__global__ __launch_bounds__(256) void k_c(const uint4* __restrict__ tab, const unsigned* __restrict__ idx, unsigned n_iter, unsigned mask, unsigned* out) {
    unsigned i = idx[blockIdx.x * 256 + threadIdx.x];
    unsigned acc = 0;
    const int lane = threadIdx.x & 63, q = lane & 3, qbase = lane & ~3;
    for (unsigned it = 0; it < n_iter; ++it) {
        unsigned want = i & mask;
        unsigned x = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned rec = __shfl(want, qbase + j);
            uint4 g = tab[(size_t)rec * 4 + q];
            x ^= g.x ^ g.y ^ g.z ^ g.w;
        }
        acc += x;
        i = i * 1664525u + 1013904223u + (x & 1u);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main(int argc, char** argv) {
    const unsigned n_threads = 256 * 256 * 6, n_iter = 64;
    for (unsigned log_recs : {14u, 18u, 21u}) {          // 1 MB, 16 MB, 128 MB tables of 64-B records
        unsigned recs = 1u << log_recs;
        uint4* tab; unsigned *idx, *out;
        hipMalloc(&tab, (size_t)recs * 64); hipMalloc(&idx, n_threads * 4); hipMalloc(&out, n_threads * 4);
        std::vector<unsigned> h(n_threads); for (auto& v : h) v = rand();
        hipMemcpy(idx, h.data(), n_threads * 4, hipMemcpyHostToDevice);
        hipMemset(tab, 1, (size_t)recs * 64);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const char* names[3] = {"A per-lane 4x16B", "B quad-coop + shfl", "C quad-coop loads only"};
        for (int v = 0; v < 3; ++v) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (v == 0) hipLaunchKernelGGL(k_a, dim3(n_threads / 256), dim3(256), 0, 0, tab, idx, n_iter, recs - 1, out);
                if (v == 1) hipLaunchKernelGGL(k_b, dim3(n_threads / 256), dim3(256), 0, 0, tab, idx, n_iter, recs - 1, out);
                if (v == 2) hipLaunchKernelGGL(k_c, dim3(n_threads / 256), dim3(256), 0, 0, tab, idx, n_iter, recs - 1, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            double bytes = (double)n_threads * n_iter * 64;
            printf("table %4u MB  %-24s %7.3f ms  %7.1f GB/s  %6.2f Grec/s\n", recs / 16384, names[v], best, bytes / best / 1e6, n_threads * (double)n_iter / best / 1e6);
        }
        hipFree(tab); hipFree(idx); hipFree(out);
    }
    return 0;
}
