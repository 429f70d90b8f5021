// End of a sample and of a pass: the sample's colour from its demodulated sums (path_tracer.rgen:105-118) and the running mean
// into the targets (write_all_outputs, shader/path_tracer.glsl:535-576; shader/gbuffer.glsl:18-28).  Device-only.
#pragma once
#include "pt_state.h"
#include "tonemap.h"

namespace tr {

namespace {

// end of one sample (path_tracer.rgen:105-118): sum_color += first_hit_material.emission + modulate_color(first_hit_material,
// diffuse, reflection) with material.glsl:57-65; sum_diffuse / sum_reflection when those targets exist
TR_DEV f4 sample_color(const PtParams& P, f4 d, f4 r, f4 fm, f4 fe) {   // rgb = emission + modulate_color(...), a = first-hit alpha
    const f3 albedo = P.opt.use_white_albedo_on_first_bounce ? F3(1) : F3(fm);
    const float metallic = fm.w;
    const float approx_fresnel = 0.02f;
    const f3 dd = F3(d) * albedo * (1 - metallic);
    const f3 rr = F3(r) * mix3(F3(approx_fresnel), albedo, metallic) / mixf(approx_fresnel, 1.0f, metallic);
    return F4(F3(fe) + (dd + rr), fe.w);
}

TR_DEV void accumulate_sample_path(const PtParams& P, const PathBuffers& pb, uint i) {   // i: path id
    u4 misc = pb.misc[i];
    if (misc.w & 1u) return;
    const f4 s = pb.sum_color[i], d = pb.diffuse[i], r = pb.reflection[i];
    const f4 c = sample_color(P, d, r, pb.first_mat[i], pb.first_emis[i]);
    pb.sum_color[i] = F4(s.x + c.x, s.y + c.y, s.z + c.z, c.w);
    if (pb.sum_diffuse) {
        const f4 sd = pb.sum_diffuse[i], sr = pb.sum_reflection[i];
        pb.sum_diffuse[i] = F4(sd.x + d.x, sd.y + d.y, sd.z + d.z, sd.w + d.w);
        pb.sum_reflection[i] = F4(sr.x + r.x, sr.y + r.y, sr.z + r.z, sr.w + r.w);
    }
}

// write_all_outputs (path_tracer.glsl:535-576) + accumulate_gbuffer_{color,diffuse,reflection} (gbuffer.glsl:18-28,68-78,118-128)
TR_DEV void resolve_path(const PtParams& P, const PathBuffers& pb, uint i) {   // i: path id
    uint lx, ly, lz;
    launch_coord(P.L, i, lx, ly, lz);
    int wx, wy;
    if (!get_write_pixel_pos(P.L, lx, ly, wx, wy)) return;
    if ((uint)wx >= P.target_w || (uint)wy >= P.target_h) return;
    const float spp = (float)P.opt.samples_per_pass;
    const size_t idx = ((size_t)lz * P.target_h + (uint)wy) * P.target_w + (uint)wx;
    const uint prev_samples = P.samples_accumulated + P.previous_samples;
    const float keep = prev_samples != 0 ? (float)prev_samples / (float)((uint)P.opt.samples_per_pass + prev_samples) : 0.0f;
    auto accumulate = [&](void* image, f4 value) {
        f4* target = reinterpret_cast<f4*>(image);
        if (prev_samples != 0) value = mix4(value, target[idx], keep);
        target[idx] = value;
        if (P.tm_display && image == P.T.color) P.tm_display[idx] = tonemap_pixel(value, P.tm_op, P.tm_exposure, P.tm_gamma, P.tm_grid, (uint)wx, (uint)wy);
    };
    if (P.fused_resolve) {
        // one sample per pass: the sums are the sample itself (0 + x and x / 1 are exact), k_accumulate_sample is not launched.
        // A launch id whose pixel is invalid never gets here: get_write_pixel_pos fails with get_pixel_pos.
        const f4 d = pb.diffuse[i], r = pb.reflection[i];
        if (P.T.color) {
            const f4 c = sample_color(P, d, r, pb.first_mat[i], pb.first_emis[i]);
            accumulate(P.T.color, F4(c.x, c.y, c.z, P.opt.transparent_background ? c.w : 1.0f));
        }
        if (P.T.diffuse) accumulate(P.T.diffuse, d);
        if (P.T.reflection) accumulate(P.T.reflection, r);
        return;
    }
    if (P.T.color) {
        const f4 s = pb.sum_color[i];
        accumulate(P.T.color, F4(s.x / spp, s.y / spp, s.z / spp, P.opt.transparent_background ? s.w : 1.0f));
    }
    if (P.T.diffuse) { const f4 s = pb.sum_diffuse[i]; accumulate(P.T.diffuse, F4(s.x / spp, s.y / spp, s.z / spp, s.w / spp)); }
    if (P.T.reflection) { const f4 s = pb.sum_reflection[i]; accumulate(P.T.reflection, F4(s.x / spp, s.y / spp, s.z / spp, s.w / spp)); }
}

__global__ __launch_bounds__(KB) void k_accumulate_sample(PtParams P, PathBuffers pb) {
    const uint i = blockIdx.x * KB + threadIdx.x;
    if (i < P.n_ids) accumulate_sample_path(P, pb, i + P.id_offset);
}
__global__ __launch_bounds__(KB) void k_resolve(PtParams P, PathBuffers pb) {
    const uint i = blockIdx.x * KB + threadIdx.x;
    if (i < P.n_ids) resolve_path(P, pb, i + P.id_offset);
}

}  // namespace

}  // namespace tr
