// Path state, launch parameters and queue appends of the wavefront path tracer: what the bounce kernels share.
// Device-only and free of host headers: this file is also compiled at run time (hipRTC) as part of a specialised shading
// program (shade_spec.hip, specialize.cc).
#pragma once
#include "shading.h"

#ifndef TR_BLOCK
#define TR_BLOCK 256
#endif

namespace tr {

constexpr int KB = TR_BLOCK;

struct PathBuffers {
    f4* org_pdf;      // origin.xyz, bsdf_pdf
    f4* dir_reg;      // direction.xyz, regularization
    f4* atten_alpha;  // attenuation.rgb, first-hit albedo.a
    f4* diffuse;      // demodulated diffuse light of the current sample (path_tracer.glsl:376), a = 1/length at bounce 1
    f4* reflection;   // demodulated reflected light, same layout
    f2* plobes;       // primary_lobes as add_demodulated_color uses them: (diffuse + transmission, dielectric + metallic reflection)
    f4* first_mat;    // first_hit_material: albedo.rgb, metallic
    f4* first_emis;   // first_hit_material.emission (= bounce-0 light), albedo.a
    u4* rng;          // random_sampler.seed
    u4* misc;         // payload.random_seed, sobol_index, launch linear id, flags (bit0 = dead)
    int4* hit;        // instance, primitive, bary.u bits, bary.v bits (u carries t for sphere lights)
    f4* sum_color;    // sum over the samples of one pass (+ alpha of the last sample)
    f4* sum_diffuse;  // only allocated when the diffuse / reflection targets are requested
    f4* sum_reflection;
    // shadow rays of the current bounce (compact)
    f4* sh_org_tmax;  // origin.xyz, tmax
    f4* sh_dir_id;    // direction.xyz, path id bits
    f4* sh_contrib;   // rgb radiance if visible, luminance for the indirect clamp
    f2* sh_lobes;     // lobe weights the contribution is demodulated with
    f4* sh_cweight;   // direct_stage only: modulate_bsdf(first hit, lobes), the weight of the sample in the colour target
    uint* queue[2];
    // per lane and bounce b (BC_STRIDE words apart): live paths entering b, shadow rays of b, work cursors of the closest-hit /
    // shadow kernel of b (BC_*).  Zeroed by k_raygen; nothing has to be rotated between bounces.
    uint* bounce;
    int* qspill;      // deep stack entries of the quad-cooperative tail of the closest-hit waves (trace_quad.h): 16 * TR_QSPILL words per wave of a launch
    uint* counters;   // per lane: statistics (CNT_*): overflow flag, ray / node / triangle / alpha / surface counts, debug slots
};

// Every counter of PathBuffers::bounce sits in its own 256 bytes: k_shade appends to the shadow queue and to the next
// bounce's queue with one atomic per wave each, and two hot words in one cache line serialise in the same L2 channel
// (measured: 0.83 ms instead of 0.56 ms for one k_shade launch of 2 M paths).
enum { BC_QUEUE = 0, BC_SHADOW = 64, BC_CUR_CLOSEST = 128, BC_CUR_SHADOW = 192, BC_STRIDE = 256 };
enum { CNT_OVERFLOW = 2, CNT_CLOSEST = 4, CNT_SHADOWRAYS = 6, CNT_NODES = 8, CNT_TRIS = 10, CNT_ALPHA = 12, CNT_SURF = 14, CNT_MAXVIS = 16, CNT_DBG = 17,
       CNT_MAXSP = 30, CNT_CNODES = 32, CNT_PH_NODE = 34, CNT_PH_TRI = 36, CNT_PH_NODE16 = 38, CNT_PH_NODE8 = 40, CNT_LV_NODE16 = 42, CNT_PH_QNODE = 44, CNT_PH_QTRI = 46, CNT_PH_HIST = 48, CNT_WORDS = 64 };   // statistics of a lane; queue lengths and work cursors live in PathBuffers::bounce

struct PtParams {
    trhip_pt_options opt;
    LaunchCtx L;
    uint viewports;
    uint n_launch;                // launch_w * launch_h * viewports
    uint id_offset, n_ids;        // the slice of path ids this launch group (lane) works on
    uint max_sobol_bounces;
    uint sample_counter, rng_seed;
    uint previous_samples;        // control.previous_samples of the pass
    uint sample_in_pass;
    uint rng_sample;              // index of this sample in the pixel's sequence: sample_base + sample_stride * (previous_samples + sample_in_pass)
    uint vp_base, vp_stride;      // local layer l renders viewport vp_base + l * vp_stride (trhip_pt_set_shard)
    uint frame_views;             // trhip_pt_set_frame_batch: layers per frame; layer l belongs to frame l / frame_views of the launch
    uint frame_counter_step;      // ... whose sample counter is sample_counter + that * frame_counter_step
    uint samples_accumulated;
    uint target_w, target_h;
    float prob_point, prob_tri, prob_dir, prob_env;   // get_nee_sampling_probabilities, scene constants
    int nee_point, nee_tri, nee_dir, nee_env;
    int count_work;
    uint bounce_words;            // size of PathBuffers::bounce for one lane
    int fused_resolve;            // samples_per_pass == 1: k_resolve forms the sample's colour itself
    trhip_pt_targets T;           // device images; null = target not requested
    f4* tm_display;               // trhip_pt_set_fused_tonemap: the last pass's k_resolve also writes tonemap(colour) here (null = off)
    int tm_op; float tm_exposure, tm_gamma; int tm_grid;
};


namespace {

TR_DEV void add64(uint* counters, int idx, uint v) {
    if (v) atomicAdd(reinterpret_cast<unsigned long long*>(counters + idx), (unsigned long long)v);
}

// wave-aggregated append: one atomic per wave, slots handed out in lane order
TR_DEV uint wave_append(uint* counter, bool pred) {
    unsigned long long mask = __ballot(pred);
    uint n = __popcll(mask);
    uint base = 0;
    int lane = threadIdx.x & 63;
    int leader = __ffsll((long long)mask) - 1;
    if (pred && lane == leader) base = atomicAdd(counter, n);
    base = __shfl(base, leader < 0 ? 0 : leader);
    uint rank = __popcll(mask & ((1ull << lane) - 1ull));
    return base + rank;
}

// Two appends per block iteration with one atomic each per *block*: 2 M paths are 32 k waves, and 32 k atomics on one word
// take longer than a shade launch should (the word's L2 channel handles them one by one).  Every thread of the block
// must call this (three __syncthreads).  Slots keep thread order within the block.
TR_DEV void block_append2(uint* counter_a, bool pred_a, uint& slot_a, uint* counter_b, bool pred_b, uint& slot_b) {
    __shared__ uint s_cnt[2][KB / 64];
    __shared__ uint s_base[2];
    const unsigned long long ma = __ballot(pred_a), mb = __ballot(pred_b);
    const uint lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) { s_cnt[0][wave] = (uint)__popcll(ma); s_cnt[1][wave] = (uint)__popcll(mb); }
    __syncthreads();
    if (threadIdx.x < 2) {
        uint total = 0;
        for (int w = 0; w < KB / 64; ++w) total += s_cnt[threadIdx.x][w];
        s_base[threadIdx.x] = total ? atomicAdd(threadIdx.x == 0 ? counter_a : counter_b, total) : 0u;
    }
    __syncthreads();
    uint off_a = s_base[0], off_b = s_base[1];
    for (uint w = 0; w < wave; ++w) { off_a += s_cnt[0][w]; off_b += s_cnt[1][w]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    slot_a = off_a + (uint)__popcll(ma & below);
    slot_b = off_b + (uint)__popcll(mb & below);
    __syncthreads();   // s_cnt / s_base are reused by the next iteration
}

// A launch can hold several consecutive frames (trhip_pt_set_frame_batch): its layers are frame-major, frame_views per frame.
TR_DEV uint global_viewport(const PtParams& P, uint lz) { return P.vp_base + (lz % P.frame_views) * P.vp_stride; }
TR_DEV uint sample_counter_of(const PtParams& P, uint lz) { return P.sample_counter + (lz / P.frame_views) * P.frame_counter_step; }

}  // namespace

}  // namespace tr
