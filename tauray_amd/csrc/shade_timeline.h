// Phase timeline of k_shade (an instrument, compiled only into a variant library:
//   make -C tauray_amd/csrc variant NAME=shadetl EXTRA=-DTR_SHADE_TIMELINE=1 FASTEXTRA=-DTR_SHADE_TIMELINE=1
//   tools/shade_timeline.py reads it out;  TR_SHADE_TIMELINE=2: the same stamps without the waits).
//
// Where the clocks of a shade wave go: the body of shade_bounce / shade_path / shade_surface is cut into segments by stamps.  A stamp
// (mode 1) first waits for every outstanding vector memory operation (`s_waitcnt vmcnt(0)`: the data a segment asked for is there,
// its stores are acknowledged), then reads s_memtime; one lane of the wave adds the clocks since the wave's previous stamp to the
// segment's sum in LDS.  The stamps are compiler barriers for memory operations, so no load is hoisted out of its segment: what a
// segment shows is its dependent fetch (or fetches) plus its arithmetic in isolation, i.e. the chain a lone wave pays for.  Mode 2
// keeps the barriers and drops the waits: the loads a segment issued are waited for where the compiler put the wait.
// A stamp costs the s_memtime round trip plus one LDS read-modify-write by one lane: ~100-150 clocks per stamp, booked to the
// segment that ends at it (segment "calibration" is two stamps back to back).
// The previous stamp of a wave lives in LDS, not in a register: stamps inside divergent branches are taken by some lanes only,
// and a per-lane register would book a branch twice (once by its lanes, once by the lanes that skipped it).
#pragma once
#ifndef TR_SHADE_TIMELINE
#define TR_SHADE_TIMELINE 0
#endif

#if TR_SHADE_TIMELINE
namespace tr {

enum { STL_ITER = 0,      // loop overhead since the previous iteration's last stamp (first iteration: since kernel entry)
       STL_ID,            // queue[qi] arrived
       STL_STATE,         // org_pdf, dir_reg, atten_alpha, hit, rng, plobes arrived
       STL_INSTANCE,      // shade_surface: span + instance transform arrived
       STL_SHADETRI,      // ... ShadeTri arrived, position / normals / uv / triangle-light pdf computed
       STL_ALBEDO,        // ... albedo factor + texture id, texture table entry, four taps
       STL_MR,            // ... metallic-roughness factor + texture
       STL_NORMALMAP,     // ... normal texture id (+ tangents, texture)
       STL_EMISSION,      // ... emission factor + texture, transmittance, ior
       STL_NOSURFACE,     // the sphere-light / environment branch of get_intersection_info
       STL_EMIT_MIS,      // emission with MIS, demodulated sums, first-hit stores
       STL_LIGHT,         // random numbers of the bounce + sample_explicit_light (light record fetch, its texture)
       STL_NEE_EVAL,      // material_bsdf_pdf of the light direction, MIS, the shadow record
       STL_BSDF,          // material_bsdf_sample + the next ray
       STL_WRITEBACK,     // (diffuse / reflection read-modify-write) + path state stores
       STL_APPEND,        // block_append2: three barriers, two atomics per block
       STL_QUEUE,         // shadow record + next-queue stores
       STL_CALIBRATION,   // two stamps back to back
       // inside STL_LIGHT (sample_explicit_light), booked in its place when the stamps below are compiled in (the sum is STL_LIGHT's)
       STL_L_SELECT,      // random numbers of the bounce, the kind of light, the lane's record fetched
       STL_L_TRI,         // triangle light: spherical-triangle sample, plane distance (+ emission texture)
       STL_L_ENV,         // environment: alias entry -> texel, direction, four envmap taps
       STL_L_DIR,         // directional light: cone sample
       STL_N = 24 };
enum { STL_BOUNCES = 8, STL_WORDS = STL_BOUNCES * 2 * STL_N };     // [bounce][sum | count][segment]

static __device__ unsigned long long g_shade_tl[STL_WORDS];       // per translation unit; shade_fast.hip's is the one read out
static __shared__ unsigned int s_stl[2 * STL_N + 16];             // sums, counts, previous stamp of each wave of the block

TR_DEV unsigned int stl_now() {
    unsigned long long t;
#if TR_SHADE_TIMELINE == 1
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
#else
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
#endif
    return (unsigned int)t;
}
TR_DEV void stl_begin() {      // kernel entry: every wave sets its previous stamp
    for (unsigned int i = threadIdx.x; i < 2u * STL_N; i += blockDim.x) s_stl[i] = 0;
    __syncthreads();
    const unsigned int now = stl_now();
    if ((threadIdx.x & 63u) == 0) s_stl[2 * STL_N + (threadIdx.x >> 6)] = now;
}
TR_DEV void stl_stamp(int seg) {
    const unsigned int now = stl_now();
    const unsigned long long m = __ballot(true);
    if ((int)(threadIdx.x & 63u) == __ffsll((long long)m) - 1) {
        unsigned int* prev = &s_stl[2 * STL_N + (threadIdx.x >> 6)];
        atomicAdd(&s_stl[seg], now - *prev);
        atomicAdd(&s_stl[STL_N + seg], 1u);
        *prev = stl_now();      // the bookkeeping itself is not booked
    }
}
TR_DEV void stl_end(int bounce) {      // kernel exit: the block's sums go to the table
    __syncthreads();
    if (bounce < 0 || bounce >= STL_BOUNCES) return;
    for (unsigned int i = threadIdx.x; i < 2u * STL_N; i += blockDim.x)
        if (s_stl[i]) atomicAdd(&g_shade_tl[(unsigned int)bounce * 2u * STL_N + i], (unsigned long long)s_stl[i]);
}

}  // namespace tr
#define STL(seg) tr::stl_stamp(tr::seg)
#else
#define STL(seg)
#endif
