// C ABI of libtrhip.so (include/trhip.h) + the small kernels around the path tracer:
// feature_stage, ray-level query hooks, stitch_stage and tonemap_stage.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "pt.h"
#include "trace.h"
#include "trace_quad.h"
#include "specialize.h"
#include "tonemap.h"
#include "../../include/tauray_image.hh"
#include "../../include/tauray_exr.hh"

namespace tr {

static thread_local std::string g_error;
int set_error(const std::string& msg) { g_error = msg; return 1; }

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return set_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

namespace {

constexpr int KB = TR_BLOCK;

// ---------------------------------------------------------------------------------------------------
// feature_stage: shader/rt_feature.rgen:21-45 + rt_feature.rchit:16-27 with FEATURE of src/feature_stage.cc:33-65
__global__ __launch_bounds__(KB) void k_feature(SceneView sv, LaunchCtx L, int feature, int projection, uint viewport, float min_ray_dist,
                                                f4 default_value, f4* target, uint target_w, uint target_h, uint* overflow_flag) {
    __shared__ int s_stack_rows[TR_STACK_WORDS];
    int* const s_stack = s_stack_rows + TR_STACK_ROW0;     // row -1 exists (LaneStack, trace.h)
    uint i = blockIdx.x * KB + threadIdx.x;
    if (i >= L.launch_w * L.launch_h) return;
    uint lx = i % L.launch_w, ly = i / L.launch_w;
    int px, py, wx, wy;
    if (!get_pixel_pos(L, lx, ly, px, py) || !get_write_pixel_pos(L, lx, ly, wx, wy)) return;
    const CameraData cam = sv.cameras[viewport];
    f3 origin, dir;
    get_screen_camera_ray(L, px, py, cam, projection, false, F2(0), F2(0.5f), origin, dir);
    f3 ray_origin = projection == 2 ? origin : F3(cam.origin);   // rt_feature.rgen:35 traces from cam.origin
    HitRecord hit;
    TraceStats st = {};
    int overflow = 0;
    trace_closest4<1, false>(sv, ray_origin, dir, min_ray_dist, __builtin_huge_valf(), false, 0u, s_stack + threadIdx.x, hit, st, overflow);
    if (overflow) *overflow_flag = 1;
    f4 data = default_value;
    if (hit.instance_id >= 0) {
        SurfacePoint v;
        SampledMaterial mat;
        shade_surface(sv, hit.instance_id, hit.primitive_id, hit.u, hit.v, dir, ray_origin, false, 0, false, v, mat);
        switch (feature) {
            default:
            case 0: data = mat.albedo; break;
            case 1: data = F4(v.mapped_normal, 1); break;
            case 2: data = F4(F3(mul(cam.view, F4(v.mapped_normal, 0))), 1); break;
            case 3: data = F4(v.pos, 1); break;
            case 4: data = mul(cam.view, F4(v.pos, 1)); break;
            case 5: data = F4(hit.t, hit.t, hit.t, 1); break;
            case 6: data = F4(v.pos - surface_prev_pos(sv, hit.instance_id, hit.primitive_id, hit.u, hit.v), 1); break;
            case 7: {
                const f3 pp = surface_prev_pos(sv, hit.instance_id, hit.primitive_id, hit.u, hit.v);
                data = F4(F3(mul(cam.view, F4(v.pos, 1)) - mul(sv.prev_cameras[viewport].view, F4(pp, 1))), 1);
                break;
            }
            case 8: data = F4(get_camera_projection(sv.prev_cameras[viewport], projection, surface_prev_pos(sv, hit.instance_id, hit.primitive_id, hit.u, hit.v)), 1); break;
            case 9: data = F4((float)hit.instance_id, (float)hit.primitive_id, 0, 1); break;
        }
    }
    if ((uint)wx >= target_w || (uint)wy >= target_h) return;
    target[(size_t)wy * target_w + (uint)wx] = data;
}

// trhip_calibrate_valu: 4096 x 64 independent v_fma_f32 per wave
__global__ __launch_bounds__(KB) void k_calibrate_fma(float* out, int iters, float a, float b) {
    float r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = (float)threadIdx.x + (float)k;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[k]) : "v"(a), "v"(b));
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += r[k];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// trhip_calibrate_l1: independent global_load_dwordx4 out of a 16 KB footprint (the same addresses for every wave: L1 hits), every
// lane in a 128-byte line of its own - 64 line (tag) accesses per wave instruction
__global__ __launch_bounds__(KB) void k_calibrate_l1(const char* base, int iters, float* out) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const char* p = base + (size_t)(threadIdx.x & 63u) * 128u;
    v4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        v4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const char* q = p + (u & 1) * 8192 + ((u >> 1) & 1) * 16;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[u]) : "v"(q) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc.x == 123.456f) out[0] = acc.y + acc.z + acc.w;
}

// ray-level hooks
// Whole waves walk the ray list together and use the wave-level traversal with its quad-cooperative tail (trace_quad.h),
// which is what the frame's closest-hit kernels run.
__global__ __launch_bounds__(KB) void k_query_closest(SceneView sv, uint n, const float* rays, const uint* seeds, int include_lights,
                                                      HitRecord* out, uint* overflow_flag, int* qspill) {
    __shared__ int s_stack_rows[TR_STACK_WORDS];
    int* const s_stack = s_stack_rows + TR_STACK_ROW0;     // row -1 exists (LaneStack, trace.h)
    __shared__ int s_owner[(KB / 64) * TR_OWNER_WORDS];
    int* my_stack = s_stack + threadIdx.x;
    QuadCtx qc;
    qc.wave_stack = s_stack + (threadIdx.x & ~63u);
    qc.owner_tab = s_owner + (threadIdx.x >> 6) * TR_OWNER_WORDS;
    qc.spill = qspill + ((size_t)blockIdx.x * (KB / 64) + (threadIdx.x >> 6)) * (16u * TR_QSPILL);
    int overflow = 0;
    TraceStats st = {};
    for (uint base = blockIdx.x * KB; base < n; base += gridDim.x * KB) {
        const uint i = base + threadIdx.x;
        const bool valid = i < n;
        float r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (valid) for (int k = 0; k < 8; ++k) r[k] = rays[(size_t)i * 8 + k];
        const uint seed = (valid && seeds) ? seeds[i] : 0u;
        HitRecord hit;
#if TR_QUAD_SWITCH > 0
        if (seeds) trace_closest_wave4<0, false>(sv, valid, F3(r[0], r[1], r[2]), F3(r[4], r[5], r[6]), r[3], r[7], include_lights != 0, seed, my_stack, qc, hit, st, overflow);
        else trace_closest_wave4<1, false>(sv, valid, F3(r[0], r[1], r[2]), F3(r[4], r[5], r[6]), r[3], r[7], include_lights != 0, 0u, my_stack, qc, hit, st, overflow);
#else
        if (valid) {
            if (seeds) trace_closest4<0, false>(sv, F3(r[0], r[1], r[2]), F3(r[4], r[5], r[6]), r[3], r[7], include_lights != 0, seed, my_stack, hit, st, overflow);
            else trace_closest4<1, false>(sv, F3(r[0], r[1], r[2]), F3(r[4], r[5], r[6]), r[3], r[7], include_lights != 0, 0u, my_stack, hit, st, overflow);
        }
#endif
        if (valid) out[i] = hit;
    }
    if (overflow) *overflow_flag = 1;
}
__global__ __launch_bounds__(KB) void k_query_shadow(SceneView sv, uint n, const float* rays, float* out, uint* overflow_flag) {
    __shared__ int s_stack_rows[TR_STACK_WORDS];
    int* const s_stack = s_stack_rows + TR_STACK_ROW0;     // row -1 exists (LaneStack, trace.h)
    int overflow = 0;
    TraceStats st = {};
    for (uint i = blockIdx.x * KB + threadIdx.x; i < n; i += gridDim.x * KB) {
        const float* r = rays + (size_t)i * 8;
        out[i] = trace_shadow4<false>(sv, F3(r[0], r[1], r[2]), F3(r[4], r[5], r[6]), r[3], r[7], s_stack + threadIdx.x, st, overflow);
    }
    if (overflow) *overflow_flag = 1;
}

// stitch_stage: shader/stitch_scanline.comp:20-50 and shader/stitch_shuffled_strips.comp:20-63 for ONE partial image
__global__ __launch_bounds__(KB) void k_stitch(LaunchCtx L, const f4* partial, uint pw, uint ph, f4* primary, uint viewports, float blend_ratio) {
    uint per_view = L.strategy == 2 ? L.launch_w : pw * ph;
    uint i = blockIdx.x * KB + threadIdx.x;
    if (i >= per_view * viewports) return;
    uint z = i / per_view, p = i % per_view;
    uint sx, sy, ox, oy;
    if (L.strategy == 2) {
        uint j = permute_region_id(L.index + p, L.size_x, L.size_y, L.count);
        if (j >= L.size_x * L.size_y) return;
        sx = p % L.size_x; sy = p / L.size_x;
        ox = j % L.size_x; oy = j / L.size_x;
    } else {
        sx = p % pw; sy = p / pw;
        ox = sx; oy = sy * L.count + L.index;
        if (oy >= L.size_y) return;
    }
    f4 c = partial[((size_t)z * ph + sy) * pw + sx];
    size_t o = ((size_t)z * L.size_y + oy) * L.size_x + ox;
    if (blend_ratio < 1.0f) c = mix4(primary[o], c, blend_ratio);
    primary[o] = c;
}

// the same for up to TR_STITCH_BATCH partial images in one launch (blockIdx.y = partial)
#define TR_STITCH_BATCH 15
struct StitchBatch { LaunchCtx L[TR_STITCH_BATCH]; const f4* partial[TR_STITCH_BATCH]; uint pw[TR_STITCH_BATCH], ph[TR_STITCH_BATCH]; };
__global__ __launch_bounds__(KB) void k_stitch_batch(StitchBatch B, f4* primary, uint viewports, float blend_ratio) {
    const uint e = blockIdx.y;
    const LaunchCtx L = B.L[e];
    const uint pw = B.pw[e], ph = B.ph[e];
    const f4* partial = B.partial[e];
    uint per_view = L.strategy == 2 ? L.launch_w : pw * ph;
    uint i = blockIdx.x * KB + threadIdx.x;
    if (i >= per_view * viewports) return;
    uint z = i / per_view, p = i % per_view;
    uint sx, sy, ox, oy;
    if (L.strategy == 2) {
        uint j = permute_region_id(L.index + p, L.size_x, L.size_y, L.count);
        if (j >= L.size_x * L.size_y) return;
        sx = p % L.size_x; sy = p / L.size_x;
        ox = j % L.size_x; oy = j / L.size_x;
    } else {
        sx = p % pw; sy = p / pw;
        ox = sx; oy = sy * L.count + L.index;
        if (oy >= L.size_y) return;
    }
    f4 c = partial[((size_t)z * ph + sy) * pw + sx];
    size_t o = ((size_t)z * L.size_y + oy) * L.size_x + ox;
    if (blend_ratio < 1.0f) c = mix4(primary[o], c, blend_ratio);
    primary[o] = c;
}

// tonemap_stage: shader/tonemap.glsl:35-55 + tonemap_{gamma,filmic,reinhard,reinhard_luminance}.comp
__global__ __launch_bounds__(KB) void k_tonemap(const f4* in, f4* out, uint w, uint h, uint layers, int op, float exposure, float gamma, int grid) {
    size_t i = (size_t)blockIdx.x * KB + threadIdx.x;
    size_t n = (size_t)w * h * layers;
    if (i >= n) return;
    out[i] = tonemap_pixel(in[i], op, exposure, gamma, grid, (uint)(i % w), (uint)((i / w) % h));
}

template <typename T>
int upload_array(T*& dst, const void* src, size_t count) {
    dst = nullptr;
    if (count == 0 || !src) return 0;
    HIPCHK(hipMalloc(&dst, count * sizeof(T)));
    HIPCHK(hipMemcpy(dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

}  // namespace
}  // namespace tr

using namespace tr;

struct trhip_device {
    int hip_device = 0;
    DeviceScene scene;
    uint* overflow_flag = nullptr;
    int* qspill = nullptr;          // deep-stack slice per wave of a k_query_closest launch (trace_quad.h)
};
constexpr uint QUERY_BLOCKS = 2048;
struct trhip_pt {
    trhip_device* dev;
    PtStage* stage;
};

#define DEVCHK(dev) do { if (!(dev)) return set_error("null trhip_device"); hipError_t e_ = hipSetDevice((dev)->hip_device); \
    if (e_ != hipSuccess) return set_error(std::string("hipSetDevice: ") + hipGetErrorString(e_)); } while (0)

extern "C" {

const char* trhip_last_error(void) { return g_error.c_str(); }

int trhip_image_decode(const void* data, size_t bytes, uint32_t* width, uint32_t* height, uint32_t* channels_in_file, uint8_t** rgba_out) {
    if (!data || !width || !height || !rgba_out) return set_error("trhip_image_decode: null argument");
    try {
        tr::image::decoded d = tr::image::decode(static_cast<const uint8_t*>(data), bytes);
        uint8_t* p = static_cast<uint8_t*>(malloc(d.rgba.size() ? d.rgba.size() : 1));
        if (!p) return set_error("trhip_image_decode: out of memory");
        memcpy(p, d.rgba.data(), d.rgba.size());
        *width = d.w; *height = d.h; *rgba_out = p;
        if (channels_in_file) *channels_in_file = (uint32_t)d.channels_in_file;
    } catch (const std::exception& e) { return set_error(e.what()); }
    return 0;
}
int trhip_image_decode_texels(const void* data, size_t bytes, uint32_t* width, uint32_t* height, uint32_t* channels_in_file, uint32_t* bits, uint8_t** texels_out) {
    if (!data || !width || !height || !bits || !texels_out) return set_error("trhip_image_decode_texels: null argument");
    try {
        tr::image::decoded d = tr::image::decode(static_cast<const uint8_t*>(data), bytes);
        const bool wide = d.bits == 16;
        const size_t n = wide ? d.rgba16.size() * 2 : d.rgba.size();
        uint8_t* p = static_cast<uint8_t*>(malloc(n ? n : 1));
        if (!p) return set_error("trhip_image_decode_texels: out of memory");
        memcpy(p, wide ? static_cast<const void*>(d.rgba16.data()) : static_cast<const void*>(d.rgba.data()), n);
        *width = d.w; *height = d.h; *bits = (uint32_t)d.bits; *texels_out = p;
        if (channels_in_file) *channels_in_file = (uint32_t)d.channels_in_file;
    } catch (const std::exception& e) { return set_error(e.what()); }
    return 0;
}
void trhip_image_free(uint8_t* rgba) { free(rgba); }

int trhip_exr_decode(const void* data, size_t bytes, uint32_t* width, uint32_t* height, uint32_t* channels, float** pixels_out) {
    if (!data || !width || !height || !channels || !pixels_out) return set_error("trhip_exr_decode: null argument");
    try {
        const tr::exr::image img = tr::exr::read(static_cast<const uint8_t*>(data), bytes);
        int n = 0;
        const std::vector<float> px = tr::exr::interleave_like_read_exr(img, n);
        float* p = static_cast<float*>(malloc(px.size() ? px.size() * 4 : 4));
        if (!p) return set_error("trhip_exr_decode: out of memory");
        memcpy(p, px.data(), px.size() * 4);
        *width = (uint32_t)img.width; *height = (uint32_t)img.height; *channels = (uint32_t)n; *pixels_out = p;
    } catch (const std::exception& e) { return set_error(e.what()); }
    return 0;
}
int trhip_exr_encode(const float* rgba, uint32_t width, uint32_t height, int alpha, int half, int compression, uint8_t** bytes_out, size_t* size_out) {
    if (!rgba || !bytes_out || !size_out || !width || !height) return set_error("trhip_exr_encode: null argument or empty image");
    try {
        const std::vector<uint8_t> b = tr::exr::encode(rgba, width, height, alpha != 0, half != 0, compression);
        uint8_t* p = static_cast<uint8_t*>(malloc(b.size()));
        if (!p) return set_error("trhip_exr_encode: out of memory");
        memcpy(p, b.data(), b.size());
        *bytes_out = p; *size_out = b.size();
    } catch (const std::exception& e) { return set_error(e.what()); }
    return 0;
}
void trhip_exr_free(void* p) { free(p); }

int trhip_device_create(int hip_device, trhip_device** out) {
    if (!out) return set_error("trhip_device_create: null out");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) return set_error(std::string("trhip_device_create: no HIP device available (") + hipGetErrorString(e) + ")");
    if (hip_device < 0 || hip_device >= count) return set_error("trhip_device_create: device index out of range");
    HIPCHK(hipSetDevice(hip_device));
    trhip_device* d = new trhip_device();
    d->hip_device = hip_device;
    HIPCHK(hipMalloc(&d->overflow_flag, 4));
    HIPCHK(hipMemset(d->overflow_flag, 0, 4));
    // the pipe of the default stream, once, while nothing of ours is in flight on this device: NULL is the stream most callers render on,
    // and a stage may not synchronise a caller's stream to find out later (stream_pool.hip; include/trhip.h "process requirements")
    { int cls = -1; (void)stream_pool_class(nullptr, &cls, true); }
    *out = d;
    return 0;
}
void trhip_device_destroy(trhip_device* dev) {
    if (!dev) return;
    (void)hipSetDevice(dev->hip_device);
    (void)hipDeviceSynchronize();
    dev->scene.free_all();
    if (dev->overflow_flag) (void)hipFree(dev->overflow_flag);
    if (dev->qspill) (void)hipFree(dev->qspill);
    delete dev;
}
int trhip_malloc(trhip_device* dev, size_t bytes, void** out) { DEVCHK(dev); HIPCHK(hipMalloc(out, bytes ? bytes : 16)); return 0; }
int trhip_free(trhip_device* dev, void* ptr) { DEVCHK(dev); if (ptr) HIPCHK(hipFree(ptr)); return 0; }
int trhip_upload(trhip_device* dev, void* dst, const void* src, size_t bytes, void* stream) {
    DEVCHK(dev); HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream)); HIPCHK(hipStreamSynchronize((hipStream_t)stream)); return 0;
}
int trhip_download(trhip_device* dev, void* dst, const void* src, size_t bytes, void* stream) {
    DEVCHK(dev); HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream)); HIPCHK(hipStreamSynchronize((hipStream_t)stream)); return 0;
}
int trhip_memset(trhip_device* dev, void* dst, int value, size_t bytes, void* stream) { DEVCHK(dev); HIPCHK(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream)); return 0; }
int trhip_sync(trhip_device* dev, void* stream) { DEVCHK(dev); HIPCHK(hipStreamSynchronize((hipStream_t)stream)); return 0; }
int trhip_stream_create(trhip_device* dev, void** stream_out) {
    DEVCHK(dev);
    if (!stream_out) return set_error("trhip_stream_create: null argument");
    hipStream_t st = nullptr;
    if (int rc = stream_pool_acquire(&st)) return rc;
    *stream_out = st;
    return 0;
}
int trhip_stream_destroy(trhip_device* dev, void* stream) {
    DEVCHK(dev);
    if (stream) { HIPCHK(hipStreamSynchronize((hipStream_t)stream)); stream_pool_release((hipStream_t)stream); }
    return 0;
}
int trhip_stream_pipe_class(trhip_device* dev, void* stream, int32_t* pipe_class_out) {
    DEVCHK(dev);
    if (!pipe_class_out) return set_error("trhip_stream_pipe_class: null argument");
    int c = -1;
    if (int rc = stream_pool_class((hipStream_t)stream, &c, true)) return rc;      // the caller asked: `stream` may be synchronised
    *pipe_class_out = c;
    return 0;
}
int trhip_device_get_info(trhip_device* dev, trhip_device_info* out) {
    DEVCHK(dev);
    if (!out) return set_error("trhip_device_get_info: null argument");
    memset(out, 0, sizeof(*out));
    out->struct_size = (uint32_t)sizeof(*out);
    out->hip_device = dev->hip_device;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev->hip_device));
    snprintf(out->name, sizeof(out->name), "%s", prop.gcnArchName);
    out->compute_units = prop.multiProcessorCount;
    if (hipDeviceGetPCIBusId(out->pci_bus_id, (int)sizeof(out->pci_bus_id), dev->hip_device) != hipSuccess) { (void)hipGetLastError(); out->pci_bus_id[0] = 0; }
    hipUUID uuid;
    if (hipDeviceGetUuid(&uuid, dev->hip_device) == hipSuccess) memcpy(out->uuid, uuid.bytes, 16); else (void)hipGetLastError();
    out->hw_queues_env = getenv("GPU_MAX_HW_QUEUES") ? atoi(getenv("GPU_MAX_HW_QUEUES")) : 0;
    int classes = 0, streams = 0;
    if (int rc = stream_pool_pipe_classes(&classes, &streams)) return rc;
    out->pipe_classes = classes; out->pool_streams = streams;
    static bool warned = false;
    if (!warned && classes > 0 && classes < 4 && getenv("TRHIP_DEBUG") && atoi(getenv("TRHIP_DEBUG")) != 0) {
        warned = true;
        fprintf(stderr, "trhip: this process reaches %d hardware pipe%s (GPU_MAX_HW_QUEUES=%s); the four lanes of a frame want four - "
                        "see include/trhip.h, \"process requirements\"\n", classes, classes == 1 ? "" : "s", getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "unset");
    }
    return 0;
}
int trhip_stream_wait(trhip_device* dev, void* stream, void* on) {
    DEVCHK(dev);
    if (stream == on) return 0;
    hipEvent_t e;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventRecord(e, (hipStream_t)on));
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, e, 0));
    HIPCHK(hipEventDestroy(e));   // released by the runtime once the recorded work has completed
    return 0;
}
int trhip_stream_wait_peer(trhip_device* dev, void* stream, trhip_device* on_dev, void* on) {
    if (!dev || !on_dev) return set_error("null trhip_device");
    if (dev->hip_device == on_dev->hip_device && stream == on) return 0;
    DEVCHK(on_dev);       // the event belongs to the device whose stream records it
    hipEvent_t e;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventRecord(e, (hipStream_t)on));
    DEVCHK(dev);
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, e, 0));
    HIPCHK(hipEventDestroy(e));
    return 0;
}
int trhip_copy_peer(trhip_device* dst_dev, void* dst, trhip_device* src_dev, const void* src, size_t bytes, void* stream) {
    if (!dst_dev || !src_dev) return set_error("null trhip_device");
    DEVCHK(src_dev);
    HIPCHK(hipMemcpyPeerAsync(dst, dst_dev->hip_device, src, src_dev->hip_device, bytes, (hipStream_t)stream));
    return 0;
}

int trhip_scene_upload(trhip_device* dev, const trhip_scene_desc* d) {
    DEVCHK(dev);
    if (!d) return set_error("trhip_scene_upload: null desc");
    HIPCHK(hipDeviceSynchronize());
    DeviceScene& s = dev->scene;
    s.free_all();
    if (d->instance_count && (!d->instances || !d->spans || !d->non_opaque)) return set_error("trhip_scene_upload: missing instance arrays");
    // validate spans against the vertex / index arrays (the reference trusts its own loader; a C ABI cannot)
    std::vector<uint> prefix(d->instance_count + 1, 0);
    const MeshSpan* spans = (const MeshSpan*)d->spans;
    const Instance* insts = (const Instance*)d->instances;
    uint tri_lights = 0;
    for (uint i = 0; i < d->instance_count; ++i) {
        const MeshSpan& sp = spans[i];
        if ((uint64_t)sp.vertex_offset + sp.vertex_count > d->vertex_count || (uint64_t)sp.index_offset + 3ull * sp.triangle_count > d->index_count)
            return set_error("trhip_scene_upload: instance span out of range");
        for (uint k = 0; k < 3 * sp.triangle_count; ++k)
            if (d->indices[sp.index_offset + k] >= sp.vertex_count) return set_error("trhip_scene_upload: vertex index out of range");
        prefix[i + 1] = prefix[i] + sp.triangle_count;
        {   // a non-finite transform would put NaN boxes into the tree (the driver's behaviour for them is undefined too)
            const float* mm = reinterpret_cast<const float*>(&insts[i].model);
            for (int k = 0; k < 16; ++k) if (!std::isfinite(mm[k])) return set_error("trhip_scene_upload: non-finite instance transform");
        }
        const Material& m = insts[i].mat;
        int texs[4] = {m.albedo_tex_id, m.metallic_roughness_tex_id, m.normal_tex_id, m.emission_tex_id};
        for (int t : texs) if (t >= (int)d->texture_count) return set_error("trhip_scene_upload: texture id out of range");
        if (insts[i].light_base_id >= 0) tri_lights = std::max(tri_lights, (uint)insts[i].light_base_id + sp.triangle_count);
    }
    {
        const Vertex* vv = (const Vertex*)d->vertices;
        for (uint i = 0; i < d->vertex_count; ++i)
            if (!(std::isfinite(vv[i].pos.x) && std::isfinite(vv[i].pos.y) && std::isfinite(vv[i].pos.z)))
                return set_error("trhip_scene_upload: non-finite vertex position");
    }
    if (upload_array(s.instances, d->instances, d->instance_count)) return 1;
    if (upload_array(s.spans, d->spans, d->instance_count)) return 1;
    {   // spans of the pre-transformed copy: every instance gets its own vertex range (instances may share a mesh)
        s.host_spans.assign(spans, spans + d->instance_count);
        s.host_world_spans = s.host_spans;
        uint64_t offset = 0;
        for (uint32_t i = 0; i < d->instance_count; ++i) { s.host_world_spans[i].vertex_offset = (uint)offset; offset += spans[i].vertex_count; }
        if (offset > 0xFFFFFFF0ull) return set_error("trhip_scene_upload: too many vertices for the pre-transformed copy");
        s.world_vertex_count = (uint)offset;
        if (upload_array(s.world_spans, s.host_world_spans.data(), s.host_world_spans.size())) return 1;
    }
    if (upload_array(s.vertices, d->vertices, d->vertex_count)) return 1;
    if (upload_array(s.indices, d->indices, d->index_count)) return 1;
    if (upload_array(s.point_lights, d->point_lights, d->point_light_count)) return 1;
    if (upload_array(s.directional_lights, d->directional_lights, d->directional_light_count)) return 1;
    if (upload_array(s.tex_infos, d->texture_infos, d->texture_count)) return 1;
    size_t texel_count = 0;
    const TextureInfo* ti = (const TextureInfo*)d->texture_infos;
    for (uint i = 0; i < d->texture_count; ++i) {
        if (ti[i].width == 0 || ti[i].height == 0) return set_error("trhip_scene_upload: empty texture");
        if (ti[i].format != TEXTURE_FORMAT_RGBA8 && ti[i].format != TEXTURE_FORMAT_RGBA16) return set_error("trhip_scene_upload: unknown texture format");
        if (ti[i].format == TEXTURE_FORMAT_RGBA16) s.wide_textures = 1;
        texel_count = std::max(texel_count, (size_t)ti[i].texel_offset + (size_t)ti[i].width * ti[i].height * (ti[i].format == TEXTURE_FORMAT_RGBA16 ? 2u : 1u));
    }
    if (upload_array(s.texels, d->texels, texel_count * 4)) return 1;
    if (upload_array(s.cameras, d->cameras, d->camera_count)) return 1;
    if (upload_array(s.non_opaque, d->non_opaque, d->instance_count)) return 1;
    {   // any-hit records: one per triangle of a non-opaque instance, laid out instance after instance (common.h AlphaTri)
        std::vector<uint> alpha_base(d->instance_count, 0xFFFFFFFFu);
        uint64_t n_alpha = 0;
        for (uint i = 0; i < d->instance_count; ++i) if (d->non_opaque[i]) { alpha_base[i] = (uint)n_alpha; n_alpha += spans[i].triangle_count; }
        if (n_alpha > 0x7FFFFFFFull) return set_error("trhip_scene_upload: more than 2^31 non-opaque triangles");
        if (upload_array(s.alpha_base, alpha_base.data(), alpha_base.size())) return 1;
        s.alpha_count = (uint)n_alpha;
        if (n_alpha) HIPCHK(hipMalloc(&s.alpha_tris, (size_t)n_alpha * sizeof(AlphaTri)));
    }
    if (upload_array(s.tri_prefix, prefix.data(), prefix.size())) return 1;
    s.environment_proj = -1;
    s.environment_factor = F4(0);
    if (d->envmap && d->envmap_width && d->envmap_height) {
        if (!d->alias_table) return set_error("trhip_scene_upload: envmap without alias table");
        size_t n = (size_t)d->envmap_width * d->envmap_height;
        if (upload_array(s.envmap, d->envmap, n)) return 1;          // f4 elements
        if (upload_array(s.alias_table, d->alias_table, n)) return 1;
        s.env_w = d->envmap_width; s.env_h = d->envmap_height;
        s.environment_proj = 0;
        s.environment_factor = F4(d->environment_factor[0], d->environment_factor[1], d->environment_factor[2], d->environment_factor[3]);
    }
    s.instance_count = d->instance_count; s.point_light_count = d->point_light_count;
    s.directional_light_count = d->directional_light_count; s.camera_count = d->camera_count; s.texture_count = d->texture_count;
    s.vertex_count = d->vertex_count; s.index_count = d->index_count; s.tri_count = prefix.back();
    s.gather_emissive_triangles = d->gather_emissive_triangles;
    s.host_tri_light_count = d->gather_emissive_triangles ? tri_lights : 0;
    if (int rc = build_shade_tris(s, -1, nullptr)) return rc;
    return 0;
}

int trhip_scene_update_lights(trhip_device* dev, const void* point_lights, uint32_t point_light_count, const void* directional_lights,
                              uint32_t directional_light_count) {
    DEVCHK(dev);
    DeviceScene& s = dev->scene;
    if (point_light_count != s.point_light_count || directional_light_count != s.directional_light_count)
        return set_error("trhip_scene_update_lights: the light counts are those of the uploaded scene (" + std::to_string(s.point_light_count) + " point, " +
                         std::to_string(s.directional_light_count) + " directional)");
    HIPCHK(hipDeviceSynchronize());
    if (point_light_count) HIPCHK(hipMemcpy(s.point_lights, point_lights, (size_t)point_light_count * sizeof(PointLight), hipMemcpyHostToDevice));
    if (directional_light_count) HIPCHK(hipMemcpy(s.directional_lights, directional_lights, (size_t)directional_light_count * sizeof(DirectionalLight), hipMemcpyHostToDevice));
    return 0;
}

int trhip_scene_update_cameras(trhip_device* dev, const void* camera_data, uint32_t count) {
    DEVCHK(dev);
    DeviceScene& s = dev->scene;
    HIPCHK(hipDeviceSynchronize());
    if (count > s.camera_count) {
        if (s.cameras) (void)hipFree(s.cameras);
        s.cameras = nullptr;
        HIPCHK(hipMalloc(&s.cameras, (size_t)count * sizeof(CameraData)));
    }
    if (count) HIPCHK(hipMemcpy(s.cameras, camera_data, (size_t)count * sizeof(CameraData), hipMemcpyHostToDevice));
    if (count != s.camera_count && s.prev_cameras) { (void)hipFree(s.prev_cameras); s.prev_cameras = nullptr; }   // falls back to the current cameras
    s.camera_count = count;
    return 0;
}
int trhip_scene_set_previous_cameras(trhip_device* dev, const void* camera_data, uint32_t count) {
    DEVCHK(dev);
    DeviceScene& s = dev->scene;
    if (count != s.camera_count) return set_error("trhip_scene_set_previous_cameras: count differs from the uploaded cameras");
    HIPCHK(hipDeviceSynchronize());
    if (!s.prev_cameras && count) HIPCHK(hipMalloc(&s.prev_cameras, (size_t)count * sizeof(CameraData)));
    if (count) HIPCHK(hipMemcpy(s.prev_cameras, camera_data, (size_t)count * sizeof(CameraData), hipMemcpyHostToDevice));
    return 0;
}
int trhip_scene_update_instances(trhip_device* dev, const void* instances, uint32_t count) {
    DEVCHK(dev);
    DeviceScene& s = dev->scene;
    if (count != s.instance_count) return set_error("trhip_scene_update_instances: count differs from the uploaded instances");
    if (count && !instances) return set_error("trhip_scene_update_instances: null instances");
    for (uint32_t i = 0; i < count; ++i) {
        const float* mm = reinterpret_cast<const float*>(&((const Instance*)instances)[i].model);
        for (int k = 0; k < 16; ++k) if (!std::isfinite(mm[k])) return set_error("trhip_scene_update_instances: non-finite instance transform");
    }
    HIPCHK(hipDeviceSynchronize());
    if (count) HIPCHK(hipMemcpy(s.instances, instances, (size_t)count * sizeof(Instance), hipMemcpyHostToDevice));
    // transforms changed: the acceleration structure, the tri lights and the pre-transformed vertex copy are stale
    // (their buffers stay: trhip_scene_build_accel refills them without allocating)
    s.accel_built = false;
    if (s.world_vertices) { (void)hipFree(s.world_vertices); s.world_vertices = nullptr; }
    return 0;
}

int trhip_scene_set_skin(trhip_device* dev, uint32_t instance, const void* source_vertices, const trhip_skin* skins, uint32_t vertex_count) {
    DEVCHK(dev);
    DeviceScene& s = dev->scene;
    if (instance >= s.instance_count) return set_error("trhip_scene_set_skin: instance out of range");
    const MeshSpan& sp = s.host_spans[instance];
    if (vertex_count != sp.vertex_count) return set_error("trhip_scene_set_skin: vertex_count differs from the instance's mesh");
    if (!skins && vertex_count) return set_error("trhip_scene_set_skin: null skins");
    HIPCHK(hipDeviceSynchronize());
    if (s.skin_slots.size() < s.instance_count) s.skin_slots.resize(s.instance_count);
    DeviceScene::SkinSlot& k = s.skin_slots[instance];
    if (k.source) { (void)hipFree(k.source); k.source = nullptr; }
    if (k.skins) { (void)hipFree(k.skins); k.skins = nullptr; }
    k.vertex_count = vertex_count;
    if (vertex_count == 0) return 0;
    HIPCHK(hipMalloc(&k.source, (size_t)vertex_count * sizeof(Vertex)));
    HIPCHK(hipMalloc(&k.skins, (size_t)vertex_count * sizeof(Skin)));
    if (source_vertices) HIPCHK(hipMemcpy(k.source, source_vertices, (size_t)vertex_count * sizeof(Vertex), hipMemcpyHostToDevice));
    else HIPCHK(hipMemcpy(k.source, s.vertices + sp.vertex_offset, (size_t)vertex_count * sizeof(Vertex), hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpy(k.skins, skins, (size_t)vertex_count * sizeof(Skin), hipMemcpyHostToDevice));
    return 0;
}
int trhip_scene_skin(trhip_device* dev, uint32_t instance, const float* joint_transforms, uint32_t joint_count) {
    DEVCHK(dev);
    DeviceScene& s = dev->scene;
    for (size_t k = 0; joint_transforms && k < (size_t)joint_count * 16; ++k)
        if (!std::isfinite(joint_transforms[k])) return set_error("trhip_scene_skin: non-finite joint transform");
    HIPCHK(hipDeviceSynchronize());   // frames in flight read the vertices
    if (int rc = skin_instance(s, instance, joint_transforms, joint_count, nullptr)) return rc;
    s.accel_built = false;
    if (s.world_vertices) { (void)hipFree(s.world_vertices); s.world_vertices = nullptr; }
    return 0;
}
int trhip_scene_get_vertices(trhip_device* dev, uint32_t instance, void* out_host, uint32_t max_count) {
    DEVCHK(dev);
    DeviceScene& s = dev->scene;
    if (instance >= s.instance_count) return set_error("trhip_scene_get_vertices: instance out of range");
    const MeshSpan& sp = s.host_spans[instance];
    const uint n = std::min(max_count, sp.vertex_count);
    HIPCHK(hipDeviceSynchronize());
    if (n) HIPCHK(hipMemcpy(out_host, s.vertices + sp.vertex_offset, (size_t)n * sizeof(Vertex), hipMemcpyDeviceToHost));
    return 0;
}

int trhip_scene_build_accel(trhip_device* dev, trhip_accel_info* out) {
    DEVCHK(dev);
    if (const char* b = getenv("TRHIP_BUILDER")) dev->scene.builder = std::string(b) == "lbvh" ? 0 : 1;
    if (const char* r = getenv("TRHIP_PLOC_RADIUS")) dev->scene.ploc_radius = std::max(1, atoi(r));
    if (const char* o = getenv("TRHIP_BVH_OPT")) dev->scene.optimise_rounds = std::max(0, atoi(o));
    if (const char* o = getenv("TRHIP_BVH_OPT_MOD")) dev->scene.optimise_modulus = std::max(1, atoi(o));
    if (const char* c = getenv("TRHIP_COLLAPSE")) dev->scene.collapse_by_cost = std::string(c) != "greedy";
    if (const char* l = getenv("TRHIP_NODE_LAYOUT")) dev->scene.dfs_layout = std::string(l) == "build" ? 0 : 1;
    return build_accel(dev->scene, nullptr, out);
}
int trhip_scene_set_build_mode(trhip_device* dev, int prefer_fast_build) {
    DEVCHK(dev);
    dev->scene.fast_build = prefer_fast_build != 0;
    return 0;
}
int trhip_scene_refit_accel(trhip_device* dev, trhip_accel_info* out) {
    DEVCHK(dev);
    return refit_accel(dev->scene, nullptr, out);
}

int trhip_scene_get_tri_lights(trhip_device* dev, void* out_host, uint32_t max_count) {
    DEVCHK(dev);
    uint n = std::min(max_count, dev->scene.tri_light_count);
    if (n) HIPCHK(hipMemcpy(out_host, dev->scene.tri_lights, (size_t)n * sizeof(TriLight), hipMemcpyDeviceToHost));
    return 0;
}

int trhip_pt_create(trhip_device* dev, const trhip_pt_options* opt, trhip_pt** out) {
    DEVCHK(dev);
    if (!opt || !out) return set_error("trhip_pt_create: null argument");
    if (opt->max_bounces < 1) return set_error("trhip_pt_create: max_bounces must be >= 1");
    if (opt->sampler < 0 || opt->sampler > 3) return set_error("trhip_pt_create: unknown sampler");
    if (opt->samples_per_pass < 1 || opt->samples_per_pixel < 1) return set_error("trhip_pt_create: sample counts must be >= 1");
    trhip_pt* p = new trhip_pt();
    p->dev = dev;
    p->stage = new PtStage(&dev->scene, *opt);
    *out = p;
    return 0;
}
int trhip_direct_create(trhip_device* dev, const trhip_pt_options* opt, trhip_pt** out) {
    if (int rc = trhip_pt_create(dev, opt, out)) return rc;
    (*out)->stage->direct = true;
    return 0;
}
void trhip_pt_destroy(trhip_pt* pt) {
    if (!pt) return;
    (void)hipSetDevice(pt->dev->hip_device);
    (void)hipDeviceSynchronize();
    delete pt->stage;
    delete pt;
}
int trhip_pt_set_distribution(trhip_pt* pt, const trhip_distribution* dist) {
    if (!pt || !dist) return set_error("trhip_pt_set_distribution: null argument");
    if (dist->strategy < 0 || dist->strategy > 2) return set_error("trhip_pt_set_distribution: unknown strategy");
    if (dist->strategy == 1 && (dist->count == 0 || dist->index >= dist->count)) return set_error("trhip_pt_set_distribution: bad scanline index/count");
    pt->stage->dist = *dist;
    return 0;
}
int trhip_pt_set_shard(trhip_pt* pt, uint32_t viewport_base, uint32_t viewport_stride, uint32_t sample_base, uint32_t sample_stride) {
    if (!pt) return set_error("trhip_pt_set_shard: null stage");
    if (viewport_stride == 0 || sample_stride == 0) return set_error("trhip_pt_set_shard: strides must be >= 1");
    if (sample_base >= sample_stride) return set_error("trhip_pt_set_shard: sample_base must be below sample_stride");
    pt->stage->shard_vp_base = viewport_base; pt->stage->shard_vp_stride = viewport_stride;
    pt->stage->shard_sample_base = sample_base; pt->stage->shard_sample_stride = sample_stride;
    return 0;
}
int trhip_pt_reset_accumulation(trhip_pt* pt, int reset_sample_counter) {
    if (!pt) return set_error("null trhip_pt");
    pt->stage->accumulated_samples = 0;
    if (reset_sample_counter) pt->stage->frame_counter = 0;
    return 0;
}
int trhip_pt_render(trhip_pt* pt, void* color_dev, uint32_t target_w, uint32_t target_h, uint32_t viewports, void* stream) {
    if (!pt) return set_error("null trhip_pt");
    DEVCHK(pt->dev);
    trhip_pt_targets t = {};
    t.color = color_dev;
    return trhip_pt_render_targets(pt, &t, target_w, target_h, viewports, stream);
}
int trhip_pt_render_targets(trhip_pt* pt, const trhip_pt_targets* targets, uint32_t target_w, uint32_t target_h, uint32_t viewports, void* stream) {
    if (!pt) return set_error("null trhip_pt");
    if (!targets) return set_error("trhip_pt_render_targets: null targets");
    DEVCHK(pt->dev);
    pt->stage->last_stream = (hipStream_t)stream;
    return pt->stage->render(*targets, target_w, target_h, viewports, (hipStream_t)stream);
}
int trhip_pt_set_frame_counter(trhip_pt* pt, uint32_t frame_counter) {
    if (!pt) return set_error("null trhip_pt");
    pt->stage->frame_counter = frame_counter;
    return 0;
}
int trhip_pt_set_frame_batch(trhip_pt* pt, uint32_t frames) {
    if (!pt) return set_error("null trhip_pt");
    if (frames == 0) return set_error("trhip_pt_set_frame_batch: at least one frame per launch");
    pt->stage->frame_batch = frames;
    return 0;
}
int trhip_pt_set_lanes(trhip_pt* pt, int lanes) {
    if (!pt) return set_error("null trhip_pt");
    if (lanes < 0) return set_error("trhip_pt_set_lanes: lanes must be >= 0");
    pt->stage->lanes = lanes;
    return 0;
}
int trhip_pt_set_frame_slots(trhip_pt* pt, int slots) {
    if (!pt) return set_error("null trhip_pt");
    if (slots < 0) return set_error("trhip_pt_set_frame_slots: slots must be >= 0");
    pt->stage->frame_slots = slots;
    return 0;
}
int trhip_pt_set_fused_tonemap(trhip_pt* pt, void* display_dev, const trhip_tonemap_info* info) {
    if (!pt) return set_error("null trhip_pt");
    if (display_dev && !info) return set_error("trhip_pt_set_fused_tonemap: null info");
    if (display_dev && pt->stage->direct) return set_error("trhip_pt_set_fused_tonemap: the direct stage resolves its samples in a kernel of its own; run trhip_tonemap");
    pt->stage->tm_display = display_dev;
    if (info) { pt->stage->tm_info = *info; if (info->op == 0) pt->stage->tm_info.gamma = 1.0f; }   // src/tonemap_stage.cc:159
    return 0;
}
int trhip_pt_get_lane_pipes(trhip_pt* pt, int32_t* lanes_out, int32_t pipe_classes_out[4]) {
    if (!pt || !lanes_out || !pipe_classes_out) return set_error("trhip_pt_get_lane_pipes: null argument");
    *lanes_out = pt->stage->last_lanes;
    for (int l = 0; l < 4; ++l) pipe_classes_out[l] = l < pt->stage->last_lanes ? pt->stage->last_lane_pipes[l] : -1;
    return 0;
}
int trhip_pt_set_shading_arithmetic(trhip_pt* pt, int ieee) {
    if (!pt) return set_error("null trhip_pt");
    pt->stage->ieee_shading = ieee ? 1 : 0;
    return 0;
}
int trhip_pt_set_specialization(trhip_pt* pt, int enable) {
    if (!pt) return set_error("null trhip_pt");
    pt->stage->specialize = enable ? 1 : 0;
    return 0;
}
int trhip_pt_precompile(const trhip_pt_options* opt, int shade_tris, int ieee, int count_work, const char* arch) {
    if (!opt) return set_error("trhip_pt_precompile: null options");
    if (is_cli_default_set(*opt) && shade_tris) return 0;      // the ahead-of-time instances of libtrhip.so
    SpecRequest rq{*opt, shade_tris != 0 && !opt->pre_transformed_vertices, ieee != 0, count_work != 0, SPEC_SHADE, false};
    std::string why;
    for (int program : {SPEC_SHADE, SPEC_RAYGEN}) {
        rq.program = program;
        if (spec_precompile(rq, arch, &why)) return set_error("trhip_pt_precompile: " + why);
    }
    return 0;
}
const char* trhip_kernel_cache_dir(void) {
    static const std::string dir = spec_cache_dir();
    return dir.c_str();
}
int trhip_pt_set_profiling(trhip_pt* pt, int count_work, int detailed_timing) {
    if (!pt) return set_error("null trhip_pt");
    pt->stage->count_work = count_work; pt->stage->detailed_timing = detailed_timing;
    return 0;
}
int trhip_pt_get_counters(trhip_pt* pt, trhip_counters* out) { if (!pt) return set_error("null trhip_pt"); DEVCHK(pt->dev); return pt->stage->get_counters(out, pt->stage->last_stream); }
int trhip_pt_reset_counters(trhip_pt* pt) { if (!pt) return set_error("null trhip_pt"); DEVCHK(pt->dev); HIPCHK(hipStreamSynchronize(pt->stage->last_stream)); return pt->stage->reset_counters(); }
int trhip_pt_get_timings(trhip_pt* pt, trhip_timings* out) { if (!pt) return set_error("null trhip_pt"); DEVCHK(pt->dev); return pt->stage->get_timings(out); }
int trhip_pt_get_program(trhip_pt* pt, trhip_program_info* out) { if (!pt) return set_error("null trhip_pt"); if (!out) return set_error("trhip_pt_get_program: null out"); DEVCHK(pt->dev); return pt->stage->get_program(out); }
int trhip_pt_get_phase_counters(trhip_pt* pt, trhip_phase_counters* out) { if (!pt) return set_error("null trhip_pt"); DEVCHK(pt->dev); return pt->stage->get_phase_counters(out, pt->stage->last_stream); }

int trhip_calibrate_valu(trhip_device* dev, float* ginst_per_s) {
    DEVCHK(dev);
    if (!ginst_per_s) return set_error("trhip_calibrate_valu: null out");
    int cus = 256;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev->hip_device));
    const int blocks = cus * 8, iters = 4096;      // eight waves per SIMD
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    float best = 0.0f;
    for (int rep = 0; rep < 4; ++rep) {            // the first launch warms up; the fastest of the rest counts
        HIPCHK(hipEventRecord(a, nullptr));
        hipLaunchKernelGGL(k_calibrate_fma, dim3(blocks), dim3(KB), 0, nullptr, reinterpret_cast<float*>(dev->overflow_flag), iters, 1.0001f, 0.5f);
        HIPCHK(hipEventRecord(b, nullptr));
        HIPCHK(hipEventSynchronize(b));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, a, b));
        const float g = (float)((double)blocks * (KB / 64) * (double)iters * 64.0 / ((double)ms * 1e6));
        if (rep > 0 && g > best) best = g;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *ginst_per_s = best;
    return 0;
}

int trhip_calibrate_l1(trhip_device* dev, float* gaccesses_per_s) {
    DEVCHK(dev);
    if (!gaccesses_per_s) return set_error("trhip_calibrate_l1: null out");
    int cus = 256;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev->hip_device));
    const int blocks = cus * 4, iters = 400;       // four waves per SIMD; two already reach the rate (tools/ubench/l1_tags.hip)
    char* buf = nullptr;
    HIPCHK(hipMalloc(&buf, 16384 + 64));
    HIPCHK(hipMemset(buf, 0, 16384 + 64));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    float best = 0.0f;
    for (int rep = 0; rep < 4; ++rep) {
        HIPCHK(hipEventRecord(a, nullptr));
        hipLaunchKernelGGL(k_calibrate_l1, dim3(blocks), dim3(KB), 0, nullptr, buf, iters, reinterpret_cast<float*>(dev->overflow_flag));
        HIPCHK(hipEventRecord(b, nullptr));
        HIPCHK(hipEventSynchronize(b));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, a, b));
        const float g = (float)((double)blocks * (KB / 64) * (double)iters * 8.0 * 64.0 / ((double)ms * 1e6));
        if (rep > 0 && g > best) best = g;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    (void)hipFree(buf);
    *gaccesses_per_s = best;
    return 0;
}

static LaunchCtx make_launch(const trhip_distribution& d) {
    LaunchCtx L{};
    uint lw, lh;
    get_ray_count(d, lw, lh);
    L.size_x = d.size_x; L.size_y = d.size_y; L.strategy = d.strategy; L.index = d.index; L.primary = d.primary;
    L.count = d.count;
    if (d.strategy == 2) { uint n = d.size_x * d.size_y, b = 31; while ((n >> b) < 128 && b > 0) b--; L.count = b; }
    L.launch_w = lw; L.launch_h = lh;
    return L;
}

int trhip_feature_render(trhip_device* dev, int feature, const trhip_distribution* dist, int projection, uint32_t viewport,
                         float min_ray_dist, const float default_value[4], void* color_dev, uint32_t target_w, uint32_t target_h, void* stream) {
    DEVCHK(dev);
    if (!dev->scene.accel_built) return set_error("trhip_feature_render: call trhip_scene_build_accel first");
    if (viewport >= dev->scene.camera_count) return set_error("trhip_feature_render: viewport out of range");
    LaunchCtx L = make_launch(*dist);
    size_t n = (size_t)L.launch_w * L.launch_h;
    if (n == 0) return 0;
    f4 dv = F4(default_value[0], default_value[1], default_value[2], default_value[3]);
    hipLaunchKernelGGL(k_feature, dim3((uint)((n + KB - 1) / KB)), dim3(KB), 0, (hipStream_t)stream, dev->scene.view(), L, feature, projection,
                       viewport, min_ray_dist, dv, (f4*)color_dev, target_w, target_h, dev->overflow_flag);
    HIPCHK(hipGetLastError());
    return 0;
}

int trhip_trace_closest(trhip_device* dev, uint32_t n, const void* rays_dev, const void* seeds_dev, int include_lights, void* hits_dev, void* stream) {
    DEVCHK(dev);
    if (!dev->scene.accel_built) return set_error("trhip_trace_closest: call trhip_scene_build_accel first");
    if (n == 0) return 0;
    uint blocks = std::min((n + KB - 1) / KB, QUERY_BLOCKS);
    if (!dev->qspill) HIPCHK(hipMalloc(&dev->qspill, (size_t)QUERY_BLOCKS * (KB / 64) * 16u * TR_QSPILL * sizeof(int)));
    hipLaunchKernelGGL(k_query_closest, dim3(blocks), dim3(KB), 0, (hipStream_t)stream, dev->scene.view(), n, (const float*)rays_dev,
                       (const uint*)seeds_dev, include_lights, (HitRecord*)hits_dev, dev->overflow_flag, dev->qspill);
    HIPCHK(hipGetLastError());
    return 0;
}
int trhip_trace_shadow(trhip_device* dev, uint32_t n, const void* rays_dev, void* visibility_dev, void* stream) {
    DEVCHK(dev);
    if (!dev->scene.accel_built) return set_error("trhip_trace_shadow: call trhip_scene_build_accel first");
    if (n == 0) return 0;
    uint blocks = std::min((n + KB - 1) / KB, 2048u);
    hipLaunchKernelGGL(k_query_shadow, dim3(blocks), dim3(KB), 0, (hipStream_t)stream, dev->scene.view(), n, (const float*)rays_dev,
                       (float*)visibility_dev, dev->overflow_flag);
    HIPCHK(hipGetLastError());
    return 0;
}

int trhip_stitch(trhip_device* dev, const trhip_distribution* pd, const void* partial_dev, uint32_t pw, uint32_t ph, void* primary_dev,
                 uint32_t viewports, float blend_ratio, void* stream) {
    DEVCHK(dev);
    if (!pd || pd->strategy == 0) return set_error("trhip_stitch: needs a scanline or shuffled-strips partial");
    LaunchCtx L = make_launch(*pd);
    size_t per_view = pd->strategy == 2 ? L.launch_w : (size_t)pw * ph;
    size_t n = per_view * viewports;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_stitch, dim3((uint)((n + KB - 1) / KB)), dim3(KB), 0, (hipStream_t)stream, L, (const f4*)partial_dev, pw, ph,
                       (f4*)primary_dev, viewports, blend_ratio);
    HIPCHK(hipGetLastError());
    return 0;
}

int trhip_stitch_batch(trhip_device* dev, uint32_t count, const trhip_distribution* pds, const void* const* partials_dev, const uint32_t* pws,
                       const uint32_t* phs, void* primary_dev, uint32_t viewports, float blend_ratio, void* stream) {
    DEVCHK(dev);
    if (count == 0) return 0;
    if (!pds || !partials_dev || !pws || !phs) return set_error("trhip_stitch_batch: null argument");
    for (uint32_t first = 0; first < count; first += TR_STITCH_BATCH) {
        const uint32_t m = std::min<uint32_t>(TR_STITCH_BATCH, count - first);
        StitchBatch B{};
        size_t n_max = 0;
        for (uint32_t k = 0; k < m; ++k) {
            const trhip_distribution& pd = pds[first + k];
            if (pd.strategy == 0) return set_error("trhip_stitch_batch: needs scanline or shuffled-strips partials");
            B.L[k] = make_launch(pd);
            B.partial[k] = (const f4*)partials_dev[first + k]; B.pw[k] = pws[first + k]; B.ph[k] = phs[first + k];
            const size_t per_view = pd.strategy == 2 ? B.L[k].launch_w : (size_t)B.pw[k] * B.ph[k];
            n_max = std::max(n_max, per_view * viewports);
        }
        if (n_max == 0) continue;
        hipLaunchKernelGGL(k_stitch_batch, dim3((uint)((n_max + KB - 1) / KB), m), dim3(KB), 0, (hipStream_t)stream, B, (f4*)primary_dev, viewports,
                           blend_ratio);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int trhip_tonemap(trhip_device* dev, const void* in_dev, void* out_dev, uint32_t width, uint32_t height, uint32_t layers,
                  const trhip_tonemap_info* info, void* stream) {
    DEVCHK(dev);
    if (!info) return set_error("trhip_tonemap: null info");
    size_t n = (size_t)width * height * layers;
    if (n == 0) return 0;
    float gamma = info->op == 0 ? 1.0f : info->gamma;   // src/tonemap_stage.cc:159
    hipLaunchKernelGGL(k_tonemap, dim3((uint)((n + KB - 1) / KB)), dim3(KB), 0, (hipStream_t)stream, (const f4*)in_dev, (f4*)out_dev, width,
                       height, layers, info->op, info->exposure, gamma, info->alpha_grid_background);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // extern "C"
