// Phase timeline of the closest-hit traversal (an instrument, compiled only into a variant library:
//   make -C tauray_amd/csrc variant NAME=timeline EXTRA=-DTR_TIMELINE=1 ;  tools/trace_timeline.py reads it out).
//
// Where the clocks of a node phase / triangle phase of trace_closest_wave4 go: every phase is bracketed by s_memtime
// stamps - phase start, all vector loads issued, `s_waitcnt vmcnt(0)` returned (data there), phase end - and the three
// intervals are summed per wave in LDS, bucketed by the number of lanes (per-lane loop) or quads (tail) that took part.
// What is left of a wave's time in the function is votes, ballots, the loop and lanes waiting for the other phase type.
// A stamp is a scalar memory instruction plus `s_waitcnt lgkmcnt(0)`; four of them per phase stretch a phase by a few
// per cent (the table prints the run's own kernel time next to the production library's).
#pragma once
#ifndef TR_TIMELINE
#define TR_TIMELINE 0
#endif

namespace tr {

// rows of the table: eight buckets each
enum { TL_LN = 0, TL_LT = 5, TL_QN = 10, TL_QT = 15,     // lane node / lane triangle / quad node / quad triangle: count, lanes, issue, wait, compute
       TL_MISC = 20,                                     // chunks, clocks in the traversal, deal clocks, ray-fetch wait, wall clock (100 MHz) in the traversal, chunk clocks with fetch and store
       TL_LN_WAIT_HIST = 21, TL_QN_WAIT_HIST = 22, TL_LT_WAIT_HIST = 23,
       TL_LN_SPLIT = 24,     // per-lane node phases, compute split: [bucket] clocks of slab tests + sort (the rest of `compute` is pushes and the pop)
       TL_LT_SPLIT = 25,     // per-lane triangle phases: [bucket] clocks of the intersection test
       TL_LT_ALPHA = 26,     // ... [bucket] clocks of the candidate / any-hit part (the rest is the pop)
       TL_LT_ALPHA_N = 27,   // ... [bucket] phases in which some lane ran the any-hit alpha test
       TL_ROWS = 28, TL_WORDS = TL_ROWS * 8 };

#if TR_TIMELINE
static __device__ unsigned long long g_timeline[TL_WORDS];   // per translation unit; path_tracer.hip's is the one read out

TR_DEV unsigned long long tl_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
TR_DEV unsigned long long tl_wall() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
TR_DEV void tl_data_arrived() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
TR_DEV int tl_wait_bin(unsigned long long w) { return w < 250 ? 0 : w < 500 ? 1 : w < 1000 ? 2 : w < 1500 ? 3 : w < 2000 ? 4 : w < 3000 ? 5 : w < 4000 ? 6 : 7; }

struct TlPhase {
    unsigned long long t0, t1, t2, ta = 0, tb = 0;
    bool alpha = false;
    TR_DEV void begin() { t0 = tl_now(); }
    TR_DEV void mark_a() { ta = tl_now(); }
    TR_DEV void mark_b() { tb = tl_now(); }
    TR_DEV void loads_issued() { t1 = tl_now(); tl_data_arrived(); t2 = tl_now(); }
    // one lane of the wave books the phase: `row` = TL_LN ..., `units` = lanes (or quads) that took part
    TR_DEV void end(uint* tl, int row, int units, int bucket, int hist_row) {
        const unsigned long long t3 = tl_now();
        const unsigned long long m = __ballot(true);
        if (tl && (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
            atomicAdd(&tl[(row + 0) * 8 + bucket], 1u); atomicAdd(&tl[(row + 1) * 8 + bucket], (uint)units);
            atomicAdd(&tl[(row + 2) * 8 + bucket], (uint)(t1 - t0)); atomicAdd(&tl[(row + 3) * 8 + bucket], (uint)(t2 - t1));
            atomicAdd(&tl[(row + 4) * 8 + bucket], (uint)(t3 - t2));
            if (hist_row >= 0) atomicAdd(&tl[hist_row * 8 + tl_wait_bin(t2 - t1)], 1u);
            if (row == TL_LN && ta) atomicAdd(&tl[TL_LN_SPLIT * 8 + bucket], (uint)(ta - t2));
            if (row == TL_LT && ta) { atomicAdd(&tl[TL_LT_SPLIT * 8 + bucket], (uint)(ta - t2)); atomicAdd(&tl[TL_LT_ALPHA * 8 + bucket], (uint)(tb - ta)); }
        }
        if (row == TL_LT && tl && __ballot(alpha) != 0 && (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&tl[TL_LT_ALPHA_N * 8 + bucket], 1u);
    }
};
TR_DEV void tl_misc(uint* tl, int slot, unsigned long long v) {
    const unsigned long long m = __ballot(true);
    if (tl && (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&tl[TL_MISC * 8 + slot], (uint)v);
}
#define TL(...) __VA_ARGS__
#else
#define TL(...)
#endif

}  // namespace tr
