// Host-side path_tracer_stage object behind trhip_pt (mirrors the state kept by rt_stage /
// rt_camera_stage / path_tracer_stage: options, distribution params, frame and sample counters).
#pragma once
#include "build.h"

namespace tr {

void get_ray_count(const trhip_distribution& d, uint& w, uint& h);

// Streams of the current device, kept for the life of the process and classified by hardware pipe (stream_pool.hip): a new stream
// for a taker whose launches must overlap those of `overlap_with` (streams of any origin, the null stream included).
int stream_pool_acquire(hipStream_t* out, const hipStream_t* overlap_with = nullptr, int n_overlap = 0);
void stream_pool_release(hipStream_t s);
// The pipe class of any stream (-1: unknown).  A stream the pool did not create is classified by the experiment the first time it is
// seen: with `blocking` the stream is synchronised for it (trhip_stream_pipe_class: the caller asked); without, only a stream that is
// idle and not capturing is probed, anything else is -1 for now.
int stream_pool_class(hipStream_t s, int* cls, bool blocking = false);
int stream_pool_pipe_classes(int* classes_out, int* streams_out);      // distinct classes this process reaches on the current device

class PtStage {
public:
    PtStage(DeviceScene* scene, const trhip_pt_options& opt);
    ~PtStage();
    int render(const trhip_pt_targets& targets, uint target_w, uint target_h, uint viewports, hipStream_t stream);
    int get_counters(trhip_counters* out, hipStream_t stream);
    int reset_counters();
    int get_timings(trhip_timings* out);
    int get_phase_counters(trhip_phase_counters* out, hipStream_t stream);
    int get_program(trhip_program_info* out);

    DeviceScene* scene;
    trhip_pt_options opt;
    trhip_distribution dist;
    uint frame_counter = 0;          // rt_stage::frame_counter (src/rt_stage.cc:81-86)
    uint accumulated_samples = 0;    // rt_camera_stage::accumulated_samples (src/rt_camera_stage.cc:100)
    int count_work = 0, detailed_timing = 0;
    // trhip_pt_set_shard: which viewports / samples of the whole job this stage renders (view and sample sharding)
    uint shard_vp_base = 0, shard_vp_stride = 1, shard_sample_base = 0, shard_sample_stride = 1;
    int frame_slots = 0;             // trhip_pt_set_frame_slots: stages rendering next to this one (0 = unknown)
    int lanes = 0;                   // trhip_pt_set_lanes: 0 = automatic
    int last_lanes = 0, last_lane_pipes[4] = {-1, -1, -1, -1};      // trhip_pt_get_lane_pipes
    hipStream_t last_main = nullptr;
    void* tm_display = nullptr;      // trhip_pt_set_fused_tonemap
    trhip_tonemap_info tm_info{};
    uint frame_batch = 1;            // trhip_pt_set_frame_batch: consecutive frames per render() call
    int ieee_shading = -1;           // trhip_pt_set_shading_arithmetic: 1 = k_shade at IEEE fp32 for every option set, 0 = Vulkan-grade arithmetic for the command-line set, -1 = TRHIP_SHADE_FAST decides
    int specialize = -1;             // trhip_pt_set_specialization: 1 = a shading program compiled for this stage's option set (hipRTC / kernel cache), 0 = the general kernels, -1 = TRHIP_SPECIALIZE decides (default on)
    bool direct = false;             // direct_stage instead of path_tracer_stage (trhip_direct_create)
    hipStream_t last_stream = nullptr;

    // which kernels shade this stage (render() and get_program() decide it the same way)
    struct Program { bool cli_set, shade_fast, wide; const struct SpecKernels *shade, *raygen; };
    Program choose_program();

private:
    struct Impl;
    Impl* impl;
    bool timing_pending = false;
    int ensure_buffers(size_t n, bool lobe_sums);
    int resolve_pending();
    void free_buffers();
};

}  // namespace tr
