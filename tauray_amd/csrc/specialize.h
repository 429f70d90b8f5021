// Shading kernels specialised for one option set at run time (specialize.cc): the stand-in for the reference's pipeline
// compilation with the options as #defines (src/path_tracer_stage.cc:30-116 -> src/rt_pipeline.cc / src/shader_source.cc).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/trhip.h"

namespace tr {

struct SpecRequest {
    trhip_pt_options opt;      // the stage's options; spec_key() says which fields an instance pins
    bool shade_tris;           // surface hits read the ShadeTri records (the scene has them and the stage does not shade pre-transformed vertices)
    bool ieee;                 // IEEE fp32 without contraction (true) or the accuracy Vulkan asks of the reference's GLSL (shade_fast.hip)
    bool count;                // the counting instances (trhip_pt_set_profiling: count_work)
    int program;               // SPEC_SHADE: trhip_spec_shade + trhip_spec_shade_last; SPEC_RAYGEN: trhip_spec_raygen (always IEEE fp32; `ieee`, `count`
                               // and the shading options are ignored)
    bool wide_textures;        // the scene holds RGBA16 textures (the two-format texel fetch, csrc/texture.h)
};
enum { SPEC_SHADE = 0, SPEC_RAYGEN = 1 };

// the option set of the reference's command line, which has ahead-of-time instances (SpecCli, shade_kernel.h)
inline bool is_cli_default_set(const trhip_pt_options& o) {
    return o.sampler == 0 && o.film == 0 && o.mis_mode == 2 && o.bounce_mode == 2 && o.tri_light_mode == 1 && o.russian_roulette_delta == 0.0f &&
           o.indirect_clamping == 0.0f && o.regularization_gamma == 0.0f && o.depth_of_field == 0 && o.hide_lights == 0 &&
           o.use_white_albedo_on_first_bounce == 0 && o.transparent_background == 0 && o.pre_transformed_vertices == 0;
}

struct SpecKernels { hipFunction_t raygen = nullptr, shade = nullptr, shade_last = nullptr; unsigned long long code_hash = 0; /* FNV-1a of the code object that was loaded */ };

// FNV-1a of the device sources embedded in this library (rtc_sources.inc) and of the hipRTC version: what tells two builds of libtrhip.so apart
unsigned long long spec_sources_hash();
unsigned long long spec_fnv1a(unsigned long long h, const void* p, size_t n);

// The pinned fields as text: what tells two instances apart (and, with the sources and the flags, names the cache file).
std::string spec_key(const SpecRequest& r);

// The kernels of `r` on the current device: from this process's table, else from the kernel cache on disk, else compiled through
// hipRTC (a few seconds) and added to both.  nullptr + *why when that is not possible (no libhiprtc, a compile error): the caller
// renders with the general kernels.
const SpecKernels* spec_kernels(const SpecRequest& r, std::string* why);

// Compiles `r` for `arch` ("gfx950") into the kernel cache unless it is there already; needs no GPU.  0 = in the cache now.
int spec_precompile(const SpecRequest& r, const char* arch, std::string* why);

std::string spec_cache_dir();   // TRHIP_KERNEL_CACHE, else kernel_cache/ next to libtrhip.so, else ~/.cache/trhip

}  // namespace tr
