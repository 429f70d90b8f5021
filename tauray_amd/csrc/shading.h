// Device-side shading math of the path tracer: sampling helpers, GGX BSDF, light sampling, camera
// rays, vertex interpolation and material fetch.  Each function names the reference shader lines it
// reproduces.  Everything operates on plain fp32 in the reference's evaluation order.
#pragma once
#include "common.h"
#include "rng.h"
#include "texture.h"
#ifndef TR_SHADE_TIMELINE
#define TR_SHADE_TIMELINE 0
#endif
#if TR_SHADE_TIMELINE
#include "shade_timeline.h"     // the phase timeline of k_shade: an instrument of variant builds only
#else
#define STL(seg)
#endif

namespace tr {

#define TR_PI 3.14159265359f                 // shader/math.glsl:4
#define TR_1_SQRT3 0.57735026918962576451f

struct SampledMaterial {                     // shader/material.glsl:24-36 (fields the path tracer reads)
    f4 albedo; float metallic, roughness; f3 emission;
    float transmittance, ior_in, ior_out, f0;
};
struct Lobes { float transmission, diffuse, dielectric_reflection, metallic_reflection; };   // bsdf_lobes

// ------------------------------------------------------------------ math.glsl
TR_DEV f3 create_tangent(f3 normal) {        // math.glsl:12-20
    f3 major;
    if (fabsf(normal.x) < TR_1_SQRT3) major = F3(1, 0, 0);
    else if (fabsf(normal.y) < TR_1_SQRT3) major = F3(0, 1, 0);
    else major = F3(0, 0, 1);
    return normalize(cross(normal, major));
}
TR_DEV m3 create_tangent_space(f3 normal) {  // math.glsl:26-31
    f3 tangent = create_tangent(normal);
    f3 bitangent = cross(normal, tangent);
    return {{tangent, bitangent, normal}};
}
TR_DEV f3 view_to_tangent_space(f3 view, const m3& tbn) {   // math.glsl:472-478
    f3 tview = mulT(-view, tbn);
    if (tview.z < 1e-5f) tview = F3(tview.x, tview.y, fmax2(tview.z, 1e-5f));
    return normalize(tview);
}
TR_DEV f2 sample_concentric_disk(f2 u) {     // math.glsl:205-218
    f2 uo = 2.0f * u - 1.0f;
    f2 a = {fabsf(uo.x), fabsf(uo.y)};
    if (a.x < 0.0001f && a.y < 0.0001f) return F2(0);
    f2 rt = (a.x > a.y) ? F2(uo.x, TR_PI / 4 * (uo.y / uo.x)) : F2(uo.y, TR_PI / 2 - TR_PI / 4 * (uo.x / uo.y));
    return rt.x * F2(tcos(rt.y), tsin(rt.y));
}
TR_DEV float sample_blackman_harris(float u) {   // math.glsl:220-228
    bool flip = u > 0.5f;
    u = flip ? 1 - u : u;
    float vx = -0.33518669f * tpow(u, 0.5f), vy = -0.51620529f * tpow(u, 0.3333333333f);
    float vz = 1.87406934f * tpow(u, 0.25f), vw = -0.66315464f * tpow(u, 0.2f);
    float s = 0.29627329f * u + vx + vy + vz + vw;
    return flip ? 1 - s : s;
}
TR_DEV f2 sample_blackman_harris_concentric_disk(f2 u) {   // math.glsl:230-241
    f2 uo = 2.0f * u - 1.0f;
    f2 a = {fabsf(uo.x), fabsf(uo.y)};
    if (a.x < 0.0001f && a.y < 0.0001f) return F2(0);
    f2 rt = (a.x > a.y) ? F2(u.x, TR_PI / 4 * (uo.y / uo.x)) : F2(u.y, TR_PI / 2 - TR_PI / 4 * (uo.x / uo.y));
    return (2.0f * sample_blackman_harris(rt.x) - 1.0f) * F2(tcos(rt.y), tsin(rt.y));
}
TR_DEV f2 sample_regular_polygon(f2 u, float angle, uint sides) {   // math.glsl:281-292
    float side = floorf(u.x * sides);
    float ux = u.x * sides;
    u.x = ux - floorf(ux);
    float side_radians = (2.0f * TR_PI) / sides;
    float a1 = side_radians * side + angle;
    float a2 = side_radians * (side + 1) + angle;
    f2 b = F2(tsin(a1), tcos(a1));
    f2 c = F2(tsin(a2), tcos(a2));
    u = u.x + u.y > 1 ? 1 - u : u;
    return b * u.x + c * u.y;
}
TR_DEV f3 sample_cosine_hemisphere(f2 u) {   // math.glsl:294-298
    f2 d = sample_concentric_disk(u);
    return F3(d.x, d.y, sqrtf(fmax2(0.0f, 1 - dot(d, d))));
}
TR_DEV float pdf_cosine_hemisphere(f3 dir) { return fmax2(dir.z, 0.0f) * (1.0f / TR_PI); }
TR_DEV f3 sample_sphere(f2 u) {              // math.glsl:305-315
    float cos_theta = 2 * u.x - 1;
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float phi = u.y * 2 * TR_PI;
    return F3(tcos(phi) * sin_theta, tsin(phi) * sin_theta, cos_theta);
}
TR_DEV f3 sample_hemisphere(f2 u) {          // math.glsl:317-327
    float cos_theta = u.x;
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float phi = u.y * 2 * TR_PI;
    return F3(tcos(phi) * sin_theta, tsin(phi) * sin_theta, cos_theta);
}
TR_DEV f3 sample_cone(f2 u, f3 dir, float cos_theta_min) {   // math.glsl:342-358
    float cos_theta = mixf(1.0f, cos_theta_min, u.x);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float phi = u.y * 2 * TR_PI;
    f3 o = mul(create_tangent_space(dir), F3(tcos(phi) * sin_theta, tsin(phi) * sin_theta, cos_theta));
    return dot(o, dir) <= cos_theta_min ? dir : o;
}
TR_DEV f3 sample_triangle_area(f2 u, f3 A, f3 B, f3 C) {     // math.glsl:360-371
    float alpha = u.x, beta = u.y;
    if (alpha + beta > 1) { alpha = 1 - alpha; beta = 1 - beta; }
    float gamma = 1 - beta - alpha;
    return alpha * A + beta * B + gamma * C;
}
TR_DEV float determinant_accurate(f3 nA, f3 nB, f3 nC) {     // math.glsl:373-382
    float div = 1.0f / sqrtf(2.0f * fabsf(nB.x) + 2.0f);
    float e = nB.x > 0 ? div : -div;
    f3 h = nB * div + F3(e, 0, 0);
    f3 a = nA - 2.0f * h * dot(h, nA);
    f3 c = nC - 2.0f * h * dot(h, nC);
    return fabsf(a.y * c.z - c.y * a.z);
}
TR_DEV f3 sample_spherical_triangle(f2 xi, f3 A, f3 B, f3 C, float& pdf) {   // math.glsl:385-419
    f3 nA = normalize(A), nB = normalize(B), nC = normalize(C);
    float dAB = dot(nA, nB), dBC = dot(nB, nC), dAC = dot(nA, nC);
    float div = 1.0f / sqrtf(2.0f * fabsf(nB.x) + 2.0f);
    float e = nB.x > 0 ? div : -div;
    f3 h = nB * div + F3(e, 0, 0);
    f3 a = nA - 2.0f * h * (dAB * div + e * nA.x);
    f3 c = nC - 2.0f * h * (dBC * div + e * nC.x);
    float G0 = fabsf(a.y * c.z - c.y * a.z);
    float G1 = dAC + dBC;
    float G2 = 1.0f + dAB;
    float solid_angle = 2.0f * atan2f(G0, G1 + G2);
    pdf = 1.0f / solid_angle;
    float chosen_split = xi.x * solid_angle * 0.5f;
    f3 r = (G0 * tcos(chosen_split) - G1 * tsin(chosen_split)) * nA + G2 * tsin(chosen_split) * nC;
    f3 Ch = 2.0f * dot(nA, r) * r / dot(r, r) - nA;
    float d = dot(Ch, nB);
    float z = 1 - xi.y + d * xi.y;
    float st = sqrtf((1.0f - z * z) / (1.0f - d * d));
    return (z - st * d) * nB + st * Ch;
}
TR_DEV float spherical_triangle_solid_angle(f3 nA, f3 nB, f3 nC) {   // math.glsl:422-429
    return 2.0f * atan2f(determinant_accurate(nA, nB, nC), 1.0f + (dot(nA, nB) + (dot(nB, nC) + dot(nA, nC))));
}
TR_DEV float triangle_area_pdf(f3 p, f3 a, f3 b, f3 c) {             // math.glsl:441-447
    f3 normal = cross(a - b, a - c);
    float p_dist2 = dot(p, p);
    return 2.0f * p_dist2 * sqrtf(p_dist2) / fabsf(dot(normal, p));
}
TR_DEV float ray_plane_intersection_dist(f3 dir, f3 A, f3 B, f3 C) { // math.glsl:450-456
    f3 pn = normalize(cross(A - B, A - C));
    float pw = dot(A, pn);
    return fabsf(pw / dot(pn, dir));
}
TR_DEV f3 get_barycentric_coords(f3 p, f3 A, f3 B, f3 C) {           // math.glsl:458-470
    f3 ba = B - A, ca = C - A, pa = p - A;
    float bb = dot(ba, ba), bc = dot(ba, ca), cc = dot(ca, ca), pb = dot(pa, ba), pc = dot(pa, ca);
    float denom = 1.0f / (bb * cc - bc * bc);
    f3 bary;
    bary.y = (cc * pb - bc * pc) * denom;
    bary.z = (bb * pc - bc * pb) * denom;
    bary.x = 1.0f - bary.y - bary.z;
    return bary;
}

// ------------------------------------------------------------------ color.glsl
TR_DEV f3 inverse_srgb_correction(f3 col) {   // color.glsl:7-12
    f3 low = col * 0.07739938f;
    f3 high = F3(tpow(fmaf(col.x, 0.94786729f, 0.05213270f), 2.4f), tpow(fmaf(col.y, 0.94786729f, 0.05213270f), 2.4f),
                 tpow(fmaf(col.z, 0.94786729f, 0.05213270f), 2.4f));
    return F3(0.04045f < col.x ? high.x : low.x, 0.04045f < col.y ? high.y : low.y, 0.04045f < col.z ? high.z : low.z);
}
TR_DEV float rgb_to_luminance(f3 col) { return dot(col, F3(0.2126f, 0.7152f, 0.0722f)); }
TR_DEV uint rgb_to_r9g9b9e5(f3 color) {        // color.glsl:19-28
    int ex, ey, ez;
    frexpf(color.x, &ex); frexpf(color.y, &ey); frexpf(color.z, &ez);
    int e = clampi(max(ex, max(ey, ez)), -16, 15);
    float sc = exp2f((float)-e) * 512.0f;
    int r = clampi((int)floorf(color.x * sc), 0, 511);
    int g = clampi((int)floorf(color.y * sc), 0, 511);
    int b = clampi((int)floorf(color.z * sc), 0, 511);
    return (uint)r | ((uint)g << 9) | ((uint)b << 18) | ((uint)(e + 16) << 27);
}
TR_DEV f3 r9g9b9e5_to_rgb(uint rgbe) {         // color.glsl:30-34
    int r = rgbe & 0x1FF, g = (rgbe >> 9) & 0x1FF, b = (rgbe >> 18) & 0x1FF, a = (rgbe >> 27) & 0x1FF;
    return F3((float)r, (float)g, (float)b) * (1.0f / 512.0f) * exp2f((float)(a - 16));
}
TR_DEV f2 unpack_half2x16(uint p) {
    return F2(__half2float(__ushort_as_half((unsigned short)(p & 0xFFFF))), __half2float(__ushort_as_half((unsigned short)(p >> 16))));
}
TR_DEV uint pack_half2x16(f2 v) {
    return (uint)__half_as_ushort(__float2half_rn(v.x)) | ((uint)__half_as_ushort(__float2half_rn(v.y)) << 16);
}

// ------------------------------------------------------------------ alias_table.glsl
TR_DEV int latlong_direction_to_pixel_id(f3 dir, int sx, int sy) {   // alias_table.glsl:22-27
    f2 uv = F2(atan2f(dir.z, dir.x) * 0.5f, asinf(-dir.y)) / TR_PI + 0.5f;
    int px = (int)(uv.x * sx + 0.5f), py = (int)(uv.y * sy + 0.5f);
    return px + py * sx;
}
TR_DEV f3 uv_to_latlong_direction(f2 uv) {     // alias_table.glsl:29-35
    uv = (uv - 0.5f) * TR_PI;
    f3 dir = F3(tcos(2.0f * uv.x), -tsin(uv.y), tsin(2.0f * uv.x));
    float s = sqrtf(1 - dir.y * dir.y);
    dir.x *= s; dir.z *= s;
    return dir;
}

// ------------------------------------------------------------------ ggx.glsl
TR_DEV float ggx_fresnel_schlick(float cos_d, float f0) { return f0 + (1.0f - f0) * tpow(fmax2(1.0f - cos_d, 0.0f), 5.0f); }
TR_DEV float ggx_fresnel(float cos_d, const SampledMaterial& mat) {         // ggx.glsl:36-49
    if (mat.ior_in > mat.ior_out) {
        float inv_eta = mat.ior_in / mat.ior_out;
        float sin_theta2 = inv_eta * inv_eta * (1.0f - cos_d * cos_d);
        if (sin_theta2 >= 1.0f) return 1.0f;
        cos_d = sqrtf(1.0f - sin_theta2);
    } else if (mat.ior_in == mat.ior_out) return 0.0f;
    return ggx_fresnel_schlick(cos_d, mat.f0);
}
TR_DEV float fresnel_importance(float cos_d, const SampledMaterial& mat) {  // ggx.glsl:54-67
    if (mat.ior_in > mat.ior_out) {
        float inv_eta = mat.ior_in / mat.ior_out;
        float sin_theta2 = inv_eta * inv_eta * (1.0f - cos_d * cos_d);
        if (sin_theta2 >= 1.0f) return 1.0f;
        cos_d = sqrtf(1.0f - sin_theta2);
    } else if (mat.ior_in == mat.ior_out) return 0.0f;
    return mat.f0 + (fmax2(1.0f - mat.roughness, mat.f0) - mat.f0) * tpow_ge0(1.0f - cos_d, 5.0f);
}
TR_DEV float ggx_masking(float v_dot_n, float v_dot_h, float a) {           // ggx.glsl:82-87
    float a2 = a * a;
    return stepf(0.0f, v_dot_n * v_dot_h) * 2.0f / (1.0f + sqrtf(1.0f + a2 / (v_dot_n * v_dot_n) - a2));
}
TR_DEV float ggx_masking_shadowing(float v_dot_n, float v_dot_h, float l_dot_n, float l_dot_h, float a) {   // ggx.glsl:90-97
    float a2 = a * a;
    return stepf(0.0f, v_dot_n * v_dot_h) * stepf(0.0f, l_dot_n * l_dot_h) * 4.0f /
           ((1.0f + sqrtf(1.0f + a2 / fmax2(v_dot_n * v_dot_n, 1e-18f) - a2)) *
            (1.0f + sqrtf(1.0f + a2 / fmax2(l_dot_n * l_dot_n, 1e-18f) - a2)));
}
TR_DEV float ggx_masking_shadowing_predivided(float v_dot_n, float v_dot_h, float l_dot_n, float l_dot_h, float a) {  // :101-109
    float a2 = a * a;
    float denom1 = fabsf(l_dot_n) * sqrtf(a2 + (1.0f - a2) * v_dot_n * v_dot_n);
    float denom2 = fabsf(v_dot_n) * sqrtf(a2 + (1.0f - a2) * l_dot_n * l_dot_n);
    return stepf(0.0f, v_dot_n * v_dot_h) * stepf(0.0f, l_dot_n * l_dot_h) * 0.5f / (denom1 + denom2);
}
TR_DEV float ggx_distribution(float h_dot_n, float a) {                     // ggx.glsl:114-119
    float a2 = a * a;
    float denom = h_dot_n * h_dot_n * (a2 - 1.0f) + 1.0f;
    return a2 / (TR_PI * denom * denom);
}
TR_DEV void ggx_brdf_inner(f3 out_dir, f3 view_dir, f3 h, float fresnel, float distribution, float cos_d,
                           const SampledMaterial& mat, Lobes& bsdf) {      // ggx.glsl:123-147
    float cos_l = out_dir.z, cos_v = view_dir.z;
    float geometry = ggx_masking_shadowing_predivided(cos_v, cos_d, cos_l, dot(out_dir, h), mat.roughness);
    float kd = (1.0f - fresnel) * (1.0f - mat.metallic) * (1.0f - mat.transmittance);
    cos_l = fmax2(cos_l, 0.0f);
    bsdf.diffuse += kd * cos_l / TR_PI;
    bsdf.dielectric_reflection += fresnel * geometry * distribution * cos_l * (1.0f - mat.metallic);
    bsdf.metallic_reflection += geometry * distribution * cos_l * mat.metallic;
}
TR_DEV f3 ggx_vndf_sample(f3 view, float roughness, float u1, float u2) {  // ggx.glsl:215-236
    f3 v = normalize(F3(roughness * view.x, roughness * view.y, view.z));
    f3 t1 = v.z < 0.9999f ? normalize(cross(v, F3(0, 0, 1))) : F3(1, 0, 0);
    f3 t2 = cross(t1, v);
    float inv_a = 1.0f + v.z;
    float a = 1.0f / inv_a;
    float r = sqrtf(u1);
    float phi = u2 < a ? u2 * inv_a * TR_PI : TR_PI + (u2 - a) / (1.0f - a) * TR_PI;
    float p1 = r * tcos(phi);
    float p2 = r * tsin(phi) * (u2 < a ? 1.0f : v.z);
    float p3 = sqrtf(fmax2(0.0f, 1.0f - p1 * p1 - p2 * p2));
    f3 n = p1 * t1 + p2 * t2 + p3 * v;
    return normalize(F3(roughness * n.x, roughness * n.y, fmax2(0.0f, n.z)));
}
TR_DEV f3 reflect3(f3 I, f3 N) { return I - 2.0f * dot(N, I) * N; }
TR_DEV f3 refract3(f3 I, f3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return F3(0.0f);
    return eta * I - (eta * d + sqrtf(k)) * N;
}
// ggx_bsdf_sample = ggx_bsdf_sample_core(eval_all_lobes = true) (ggx.glsl:240-388)
TR_DEV void ggx_bsdf_sample(f4 ur, f3 view_dir, const SampledMaterial& mat, f3& out_dir, Lobes& bsdf, float& pdf) {
    const bool zero_roughness = mat.roughness < 0.001f;
    f3 h = zero_roughness ? F3(0, 0, 1) : ggx_vndf_sample(view_dir, mat.roughness, ur.x, ur.y);
    float cos_d = dot(view_dir, h);
    float fresnel = ggx_fresnel(cos_d, mat);
    float cos_v = view_dir.z;
    float max_albedo = fmax2(mat.albedo.x, fmax2(mat.albedo.y, mat.albedo.z));
    float specular_cutoff = mixf(1.0f, fresnel_importance(view_dir.z, mat), (1 - mat.metallic) * max_albedo);
    float diffuse_cutoff = 1.0f - mat.transmittance;
    float specular_probability = specular_cutoff;
    float diffuse_probability = (1.0f - specular_cutoff) * diffuse_cutoff;
    float transmissive_probability = (1.0f - specular_cutoff) * (1.0f - diffuse_cutoff);
    float u = ur.z;
    pdf = 0.0f;
    out_dir = F3(0);
    if (u <= specular_cutoff) {
        out_dir = reflect3(-view_dir, h);
        float cos_l = out_dir.z, cos_h = h.z;
        float G1 = ggx_masking(cos_v, cos_d, mat.roughness);
        float D = zero_roughness ? 4 * cos_l * cos_v : ggx_distribution(cos_h, mat.roughness);
        pdf = G1 * D / (4 * fabsf(cos_v)) * specular_probability;
        float diffuse_pdf = (zero_roughness ? 0 : pdf_cosine_hemisphere(out_dir) * diffuse_probability);
        pdf += diffuse_pdf;
        ggx_brdf_inner(out_dir, view_dir, h, fresnel, D, cos_d, mat, bsdf);
        if (zero_roughness) {
            bsdf.diffuse = 0;
            bsdf.dielectric_reflection /= pdf;
            bsdf.metallic_reflection /= pdf;
            pdf = 0;
        }
    } else {
        u = clampf((u - specular_cutoff) / (1 - specular_cutoff), 0.0f, 0.99999f);
        if (u <= diffuse_cutoff) {
            u = clampf(u / diffuse_cutoff, 0.0f, 0.99999f);
            out_dir = sample_cosine_hemisphere(F2(u, ur.w));
            h = normalize(view_dir + out_dir);
            float cos_h = h.z;
            cos_d = dot(view_dir, h);
            fresnel = ggx_fresnel_schlick(cos_d, mat.f0);   // ggx_fresnel_refl
            float G1 = ggx_masking(cos_v, cos_d, mat.roughness);
            float D = (zero_roughness ? 0 : ggx_distribution(cos_h, mat.roughness));
            pdf = pdf_cosine_hemisphere(out_dir) * diffuse_probability;
            float specular_pdf = G1 * D / (4 * fabsf(cos_v)) * specular_probability;
            pdf += specular_pdf;
            ggx_brdf_inner(out_dir, view_dir, h, fresnel, D, cos_d, mat, bsdf);
            if (zero_roughness) { bsdf.dielectric_reflection = 0; bsdf.metallic_reflection = 0; }
        } else {
            out_dir = normalize(refract3(-view_dir, h, mat.ior_in / mat.ior_out));
            if (any_nan(out_dir)) { out_dir = F3(0); pdf = 0; return; }
            float cos_l = out_dir.z, cos_h = h.z;
            float cos_o = dot(out_dir, h);
            float G2 = ggx_masking_shadowing(cos_v, cos_d, cos_l, cos_o, mat.roughness);
            float G1 = ggx_masking(cos_v, cos_d, mat.roughness);
            float D = zero_roughness ? 4 * cos_l * cos_v : ggx_distribution(cos_h, mat.roughness);
            float denom = mat.ior_in / mat.ior_out * cos_d + cos_o;
            bsdf.transmission += fabsf(cos_d * cos_o) * mat.transmittance * (1.0f - mat.metallic) * (1.0f - fresnel) * G2 * D /
                                 (denom * denom * fabsf(cos_v));
            pdf = (fabsf(cos_d * cos_o) * G1 * D) / (denom * denom * fabsf(cos_v)) * transmissive_probability;
            if (zero_roughness) { bsdf.transmission /= pdf; pdf = 0; }
        }
    }
}
// ggx_bsdf_pdf = ggx_bsdf_lobe_pdf(MATERIAL_LOBE_ALL) (ggx.glsl:403-510)
TR_DEV float ggx_bsdf_pdf(f3 out_dir, f3 view_dir, const SampledMaterial& mat, Lobes& bsdf) {
    float cos_l = out_dir.z, cos_v = view_dir.z;
    f3 h;
    if (cos_l > 0) h = normalize(view_dir + out_dir);
    else h = (mat.ior_in > mat.ior_out ? 1.0f : -1.0f) * normalize(mat.ior_out * out_dir + mat.ior_in * view_dir);
    float cos_h = h.z;
    float cos_d = dot(view_dir, h);
    float cos_o = dot(out_dir, h);
    float fresnel = ggx_fresnel(cos_d, mat);
    float geometry = ggx_masking_shadowing_predivided(cos_v, cos_d, cos_l, cos_o, mat.roughness);
    const bool zero_roughness = mat.roughness < 0.001f;
    float distribution = zero_roughness ? 0 : ggx_distribution(cos_h, mat.roughness);
    float max_albedo = fmax2(mat.albedo.x, fmax2(mat.albedo.y, mat.albedo.z));
    float specular_cutoff = mixf(1.0f, fresnel_importance(view_dir.z, mat), (1 - mat.metallic) * max_albedo);
    float diffuse_cutoff = 1.0f - mat.transmittance;
    float specular_probability = specular_cutoff;
    float diffuse_probability = (1.0f - specular_cutoff) * diffuse_cutoff;
    float transmissive_probability = (1.0f - specular_cutoff) * (1.0f - diffuse_cutoff);
    float G1 = ggx_masking(cos_v, cos_d, mat.roughness);
    float pdf = 0.0f;
    if (cos_l > 0) {
        float kd = (1.0f - fresnel) * (1.0f - mat.metallic) * (1.0f - mat.transmittance);
        float diffuse_pdf = pdf_cosine_hemisphere(out_dir) * diffuse_probability;
        if (!isnan(diffuse_pdf) && !isinf(diffuse_pdf) && diffuse_pdf > 0.0f) {
            bsdf.diffuse += kd * cos_l / TR_PI;
            pdf += diffuse_pdf;
        }
        float specular_pdf = G1 * distribution / (4 * fabsf(cos_v)) * specular_probability;
        if (!isnan(specular_pdf) && !isinf(specular_pdf) && specular_pdf > 0.0f) {
            bsdf.dielectric_reflection += fresnel * geometry * distribution * cos_l * (1.0f - mat.metallic);
            bsdf.metallic_reflection += geometry * distribution * cos_l * mat.metallic;
            pdf += specular_pdf;
        }
    } else {
        float denom = mat.ior_in / mat.ior_out * cos_d + cos_o;
        geometry *= 4.0f;
        float transmit_pdf = (fabsf(cos_d * cos_o) * G1 * distribution) / (fabsf(cos_v) * denom * denom * TR_PI) * transmissive_probability;
        if (!isnan(transmit_pdf) && !isinf(transmit_pdf) && transmit_pdf > 0.0f) {
            bsdf.transmission += -cos_l * fabsf(cos_d * cos_o) * mat.transmittance * (1.0f - mat.metallic) * (1.0f - fresnel) * geometry * distribution / (denom * denom);
            pdf += transmit_pdf;
        }
    }
    return pdf;
}
// material_bsdf_sample / material_bsdf_pdf (ggx.glsl:512-552)
TR_DEV void material_bsdf_sample(int bounce_mode, f4 ur, f3 view_dir, const SampledMaterial& mat, f3& out_dir, Lobes& bsdf, float& pdf) {
    if (bounce_mode == 0) {
        if (mat.transmittance > 0.0f) { out_dir = sample_sphere(F2(ur.x, ur.y)); pdf = 0.25f / TR_PI; }
        else { out_dir = sample_hemisphere(F2(ur.x, ur.y)); pdf = 0.5f / TR_PI; }
        ggx_bsdf_pdf(out_dir, view_dir, mat, bsdf);
    } else if (bounce_mode == 1) {
        float split = mat.transmittance * 0.5f;
        out_dir = (ur.z < split ? -1.0f : 1.0f) * sample_cosine_hemisphere(F2(ur.x, ur.y));
        pdf = fabsf(out_dir.z / TR_PI) * (ur.z < split ? split : 1.0f - split);
        ggx_bsdf_pdf(out_dir, view_dir, mat, bsdf);
    } else ggx_bsdf_sample(ur, view_dir, mat, out_dir, bsdf, pdf);
}
TR_DEV float material_bsdf_pdf(int bounce_mode, f3 out_dir, f3 view_dir, const SampledMaterial& mat, Lobes& bsdf) {
    if (bounce_mode == 0) {
        ggx_bsdf_pdf(out_dir, view_dir, mat, bsdf);
        if (mat.transmittance == 0 && out_dir.z <= 0) return 0.0f;
        return mat.transmittance > 0.0f ? 0.25f / TR_PI : 0.5f / TR_PI;
    } else if (bounce_mode == 1) {
        ggx_bsdf_pdf(out_dir, view_dir, mat, bsdf);
        if (mat.transmittance == 0 && out_dir.z <= 0) return 0.0f;
        float split = mat.transmittance * 0.5f;
        return fabsf(out_dir.z / TR_PI) * (out_dir.z < 0 ? split : 1.0f - split);
    }
    return ggx_bsdf_pdf(out_dir, view_dir, mat, bsdf);
}
TR_DEV f3 modulate_bsdf(const SampledMaterial& mat, const Lobes& b) {      // material.glsl:52-55
    return F3(mat.albedo) * (b.metallic_reflection + b.transmission + b.diffuse) + b.dielectric_reflection;
}

// ------------------------------------------------------------------ light.glsl
TR_DEV float get_spotlight_intensity(const PointLight& l, f3 dir) {        // light.glsl:45-58
    if (l.dir_falloff > 0) {
        float cutoff = dot(dir, -l.dir);
        cutoff = cutoff > l.dir_cutoff ? 1.0f - tpow(fmax2(1.0f - cutoff, 0.0f) / (1.0f - l.dir_cutoff), l.dir_falloff) : 0.0f;
        return cutoff;
    }
    return 1.0f;
}
TR_DEV void sample_point_light(const PointLight& pl, f2 u, f3 pos, f3& out_dir, float& out_length, f3& color, float& pdf) {  // :76-106
    f3 dir = pos - pl.pos;
    float dist2 = dot(dir, dir);
    float k = 1.0f - pl.radius * pl.radius / dist2;
    float dir_cutoff = k > 0 ? sqrtf(k) : -1.0f;
    out_dir = sample_cone(u, -normalize(dir), dir_cutoff);
    float b = dot(dir, out_dir);
    out_length = -b - sqrtf(fmax2(b * b - dist2 + pl.radius * pl.radius, 0.0f));
    color = get_spotlight_intensity(pl, normalize(-dir)) * pl.color;
    if (pl.radius == 0.0f) pdf = -dist2;
    else { color = color / (pl.radius * pl.radius * TR_PI); pdf = 1 / (2.0f * TR_PI * (1.0f - dir_cutoff)); }
}
TR_DEV float sample_point_light_pdf(const PointLight& pl, f3 pos) {        // light.glsl:108-117
    f3 dir = pos - pl.pos;
    float dist2 = dot(dir, dir);
    float k = 1.0f - pl.radius * pl.radius / dist2;
    float dir_cutoff = k > 0 ? sqrtf(k) : -1.0f;
    if (pl.radius == 0.0f) return 0;
    return 1 / (2.0f * TR_PI * (1.0f - dir_cutoff));
}
TR_DEV float sample_directional_light_pdf(const DirectionalLight& dl) {    // light.glsl:131-134
    return dl.dir_cutoff >= 1.0f ? 0.0f : 1.0f / (2.0f * TR_PI * (1.0f - dl.dir_cutoff));
}
TR_DEV float sample_triangle_light_pdf(int mode, f3 P, f3 A, f3 B, f3 C) { // light.glsl:136-179
    if (mode == 0) return triangle_area_pdf(P, A, B, C);
    float solid_angle = spherical_triangle_solid_angle(normalize(A), normalize(B), normalize(C));
    if (mode == 1) return 1.0f / solid_angle;
    return solid_angle > 1e-6f ? 1.0f / solid_angle : triangle_area_pdf(P, A, B, C);
}
TR_DEV f3 sample_triangle_light(int mode, f2 u, f3 A, f3 B, f3 C, float& pdf) {
    if (mode == 1) return sample_spherical_triangle(u, A, B, C, pdf);
    if (mode == 2) {
        float solid_angle = spherical_triangle_solid_angle(normalize(A), normalize(B), normalize(C));
        if (solid_angle > 1e-6f) return sample_spherical_triangle(u, A, B, C, pdf);
    }
    f3 P = sample_triangle_area(u, A, B, C);
    pdf = triangle_area_pdf(P, A, B, C);
    return normalize(P);
}

// ------------------------------------------------------------------ camera.glsl:25-59,99-124
TR_DEV void get_camera_ray(const CameraData& cam, int projection, bool dof, f2 pixel_coord, f2 screen_size, f2 dof_u,
                           f3& origin, f3& dir) {
    f2 uv = pixel_coord / screen_size;
    if (projection == 2) {
        // equirectangular_camera_data_buffer {view, view_inverse, origin, fov}: first 152 bytes of the slot
        const float* raw = reinterpret_cast<const float*>(&cam);
        f2 fov = F2(raw[36], raw[37]);
        uv = (uv * 2.0f - 1.0f) * fov;
        f2 c = F2(tcos(uv.x), tcos(uv.y)), s = F2(tsin(uv.x), tsin(uv.y));
        f3 t = F3(s.x * c.y, s.y, -c.x * c.y);
        dir = normalize(F3(mul(cam.view_inverse, F4(t, 0))));
        origin = F3(raw[32], raw[33], raw[34]);
        return;
    }
    uv = uv * 2.0f - 1.0f;
    if (projection == 0) {
        if (dof) {
            f2 ao = cam.dof_params.w == 0 ? sample_concentric_disk(dof_u) : sample_regular_polygon(dof_u, cam.dof_params.z, (uint)cam.dof_params.w);
            f3 view_origin = F3(ao.x * cam.dof_params.y, ao.y * cam.dof_params.y, 0);
            f3 view_dir = F3(mul(cam.proj_inverse, F4(uv.x, uv.y, 1, 1))) * cam.dof_params.x;
            view_dir = normalize(view_dir - view_origin);
            origin = F3(mul(cam.view_inverse, F4(view_origin, 1.0f)));
            dir = normalize(F3(mul(cam.view_inverse, F4(view_dir, 0))));
        } else {
            origin = F3(cam.origin);
            f4 t = mul(cam.proj_inverse, F4(uv.x, uv.y, 1, 1));
            dir = normalize(F3(mul(cam.view_inverse, F4(t.x, t.y, t.z, 0))));
        }
    } else {
        origin = F3(mul(cam.view_inverse, mul(cam.proj_inverse, F4(uv.x, uv.y, 0, 1))));
        dir = normalize(F3(mul(cam.view_inverse, F4(0, 0, -1, 0))));
    }
}

// ------------------------------------------------------------------ rt.glsl:170-231 distribution
struct LaunchCtx {
    uint size_x, size_y;          // distribution.size
    int strategy;
    uint index, count, primary;   // count = b (strip bits) for shuffled strips, as uploaded by rt_camera_stage.cc:86-87
    uint launch_w, launch_h;      // gl_LaunchSizeEXT.xy
    uint tile_lanes, tiles_per_lane;   // > 1: lane l's contiguous id range covers the tiles l, l + lanes, l + 2 lanes, ... of the image
};
// Path id -> launch coordinate.  Path ids walk the launch grid in 8x8 tiles (when the width divides by 8) instead of
// rows, so the 64 rays of a wave start from an 8x8 pixel block: more coherent traversal and texture access on every
// bounce.  Results do not depend on this mapping (all random streams are keyed by the absolute pixel).
TR_DEV void launch_coord(const LaunchCtx& L, uint i, uint& lx, uint& ly, uint& lz) {
    const uint per_layer = L.launch_w * L.launch_h;
    lz = i / per_layer;
    const uint j = i - lz * per_layer;
    if ((L.launch_w & 7u) == 0u && j < L.launch_w * (L.launch_h & ~7u)) {   // rows past the last full tile row stay row-major
        uint tile = j >> 6;
        const uint k = j & 63u, tiles_x = L.launch_w >> 3;
        if (L.tile_lanes > 1u) {   // the lanes of a frame (PtStage::render) take interleaved tiles: equal shares of sky, walls and clutter
            const uint lane = tile / L.tiles_per_lane;
            tile = (tile - lane * L.tiles_per_lane) * L.tile_lanes + lane;
        }
        const uint ty = tile / tiles_x, tx = tile - ty * tiles_x;
        lx = (tx << 3) + (k & 7u);
        ly = (ty << 3) + (k >> 3);
    } else {
        ly = j / L.launch_w;
        lx = j - ly * L.launch_w;
    }
}
TR_HD uint permute_region_id(uint i, uint size_x, uint size_y, uint b) {
    uint region_size = ((size_x * size_y) + (1u << b) - 1) >> b;
    uint region_id = i / region_size;
    uint k = b == 0 ? 0 : (bitrev32(region_id) >> (32 - b));
    return k * region_size + i % region_size;
}
TR_DEV bool get_pixel_pos(const LaunchCtx& L, uint lx, uint ly, int& px, int& py) {
    if (L.strategy == 0) { px = (int)lx; py = (int)ly; return true; }
    if (L.strategy == 1) { px = (int)lx; py = (int)(ly * L.count + L.index); return true; }
    uint j = permute_region_id(L.index + lx, L.size_x, L.size_y, L.count);
    if (j < L.size_x * L.size_y) { px = (int)(j % L.size_x); py = (int)(j / L.size_x); return true; }
    return false;
}
TR_DEV bool get_write_pixel_pos(const LaunchCtx& L, uint lx, uint ly, int& wx, int& wy) {
    if (L.strategy == 0) { wx = (int)lx; wy = (int)ly; return true; }
    if (L.strategy == 1) { wx = (int)lx; wy = (int)(L.primary == 1 ? ly * L.count + L.index : ly); return true; }
    uint j = permute_region_id(L.index + lx, L.size_x, L.size_y, L.count);
    if (L.primary == 1) { wx = (int)(j % L.size_x); wy = (int)(j / L.size_x); }
    else { wx = (int)(lx % L.size_x); wy = (int)(lx / L.size_x); }
    return j < L.size_x * L.size_y;
}
TR_DEV void get_screen_camera_ray(const LaunchCtx& L, int px, int py, const CameraData& cam, int projection, bool dof,
                                  f2 pixel_offset, f2 dof_u, f3& origin, f3& dir) {   // rt.glsl:234-241
    f2 p = F2((float)px, (float)py) + (pixel_offset * 0.5f + 0.5f);
    uint sx = L.strategy == 0 ? L.launch_w : L.size_x, sy = L.strategy == 0 ? L.launch_h : L.size_y;
    p.y = (float)sy - p.y;
    get_camera_ray(cam, projection, dof, p, F2((float)sx, (float)sy), dof_u, origin, dir);
}

// ------------------------------------------------------------------ rt.glsl:27-101, scene.glsl:87-152
struct SurfacePoint {
    f3 pos, hard_normal, smooth_normal, mapped_normal;
    float tri_light_pdf;
};

// CALC_PREV_VERTEX_POS (shader/rt.glsl:73-79): the hit point under the previous frame's model matrix.  Interpolates the
// uploaded model-space vertices (also under PRE_TRANSFORMED_VERTICES, where the reference goes through inverse(model)).
TR_DEV f3 surface_prev_pos(const SceneView& sv, int instance_id, int primitive_id, float bu, float bv) {
    const MeshSpan span = sv.obj_spans[instance_id];
    const uint* ix = sv.indices + span.index_offset + 3u * (uint)primitive_id;
    const Vertex* vb = sv.obj_vertices + span.vertex_offset;
    const f3 b = F3(1.0f - bu - bv, bu, bv);
    const f3 object_pos = vb[ix[0]].pos * b.x + vb[ix[1]].pos * b.y + vb[ix[2]].pos * b.z;
    return F3(mul(sv.instances[instance_id].model_prev, F4(object_pos, 1)));
}

// get_camera_projection (shader/camera.glsl:61-67 perspective / orthographic, :126-134 equirectangular)
TR_DEV f3 get_camera_projection(const CameraData& cam, int projection, f3 world_pos) {
    if (projection == 2) {
        f3 t = F3(mul(cam.view, F4(world_pos, 1.0f)));
        const float t_len = length(t);
        t = t / t_len;
        const float* raw = reinterpret_cast<const float*>(&cam);   // equirect layout: view, view_inverse, origin, fov
        return F3((atan2f(t.x, -t.z) / raw[36]) * 0.5f + 0.5f, (asinf(t.y) / raw[37]) * 0.5f + 0.5f, t_len);
    }
    const f4 p = mul(cam.view_proj, F4(world_pos, 1.0f));
    return F3((p.x / p.w) * 0.5f + 0.5f, (p.y / p.w) * 0.5f + 0.5f, p.w);
}

// get_interpolated_vertex + sample_material fused: fetches 3 indices, 3 x 48-byte vertices and the 288-byte
// instance once.  `want_tri_pdf` = NEE_SAMPLE_EMISSIVE_TRIANGLES.
// `rec` (a compile-time constant at every call): the vertices come from the triangle's ShadeTri record - positions, normals and texture
// coordinates from one cache line, one dependent fetch earlier than through the indices - and the tangents, which only a normal map
// reads, from SceneView::shade_tangents when one is there.  Only without `pre` (the records hold model-space vertices).
TR_DEV void shade_surface(const SceneView& sv, int instance_id, int primitive_id, float bu, float bv, f3 view, f3 ray_origin,
                          bool want_tri_pdf, int tri_light_mode, bool pre, SurfacePoint& sp, SampledMaterial& res, bool rec = false) {
    const Instance& o = sv.instances[instance_id];
    const MeshSpan span = sv.spans[instance_id];
    STL(STL_INSTANCE);
    f3 q0, q1, q2, n0, n1, n2;
    f2 t0, t1, t2;
    f4 g0 = F4(0, 0, 0, 0), g1 = g0, g2 = g0;      // tangents: with the vertices (no records), or fetched where they are used
    const uint record = span.index_offset / 3u + (uint)primitive_id;
    if (rec) {
        const ShadeTri* t = sv.shade_tris + record;
        q0 = t->pos[0]; q1 = t->pos[1]; q2 = t->pos[2];
        n0 = t->normal[0]; n1 = t->normal[1]; n2 = t->normal[2];
        t0 = t->uv[0]; t1 = t->uv[1]; t2 = t->uv[2];
    } else {
        const uint* ix = sv.indices + span.index_offset + 3u * (uint)primitive_id;
        const Vertex* vb = sv.vertices + span.vertex_offset;
        const Vertex v0 = vb[ix[0]], v1 = vb[ix[1]], v2 = vb[ix[2]];
        q0 = v0.pos; q1 = v1.pos; q2 = v2.pos;
        n0 = v0.normal; n1 = v1.normal; n2 = v2.normal;
        t0 = v0.uv; t1 = v1.uv; t2 = v2.uv;
        g0 = v0.tangent; g1 = v1.tangent; g2 = v2.tangent;
    }
    const m4 model = o.model;
    const m3 mn = upper3(o.model_normal);
    const f3 b = F3(1.0f - bu - bv, bu, bv);
    f4 model_pos = F4(q0 * b.x + q1 * b.y + q2 * b.z, 1);
    sp.pos = pre ? F3(model_pos) : F3(mul(model, model_pos));   // `pre` = PRE_TRANSFORMED_VERTICES (rt.glsl:18-22): TRANSFORM_MAT is the identity
    sp.tri_light_pdf = 0.0f;
    if (want_tri_pdf && o.light_base_id >= 0) {
        f3 p0 = pre ? q0 : transform_point(model, q0), p1 = pre ? q1 : transform_point(model, q1), p2 = pre ? q2 : transform_point(model, q2);
        sp.tri_light_pdf = sample_triangle_light_pdf(tri_light_mode, sp.pos - ray_origin, p0 - ray_origin, p1 - ray_origin, p2 - ray_origin);
    }
    const f3 sn = n0 * b.x + n1 * b.y + n2 * b.z;
    f3 smooth_normal = normalize(pre ? sn : mul(mn, sn));
    f2 uv = t0 * b.x + t1 * b.y + t2 * b.z;
    const f3 hn = cross(q1 - q0, q2 - q0);
    f3 hard_normal = normalize(pre ? hn : mul(mn, hn));
    bool back_facing = dot(hard_normal, view) > 0;
    if (back_facing) { smooth_normal = -smooth_normal; hard_normal = -hard_normal; }
    sp.hard_normal = hard_normal;
    sp.smooth_normal = smooth_normal;
    sp.mapped_normal = smooth_normal;
    STL(STL_SHADETRI);

    const Material& mat = o.mat;
    res.albedo = mat.albedo_factor;
    if (mat.albedo_tex_id >= 0) {
        f4 tex_col = sample_texture(sv, mat.albedo_tex_id, uv);
        f3 lin = inverse_srgb_correction(F3(tex_col));
        res.albedo = res.albedo * F4(lin, tex_col.w);
    }
    STL(STL_ALBEDO);
    f2 mr = F2(mat.metallic_roughness_factor.x, mat.metallic_roughness_factor.y);
    if (mat.metallic_roughness_tex_id >= 0) {
        f4 t = sample_texture(sv, mat.metallic_roughness_tex_id, uv);
        mr = mr * F2(t.z, t.y);
    }
    res.metallic = mr.x;
    res.roughness = mr.y * mr.y;
    STL(STL_MR);
    if (mat.normal_tex_id >= 0) {
        // the bitangent comes from the interpolated normal BEFORE the back-face flip (the order the vertex is assembled in, and the
        // oracle's): `front` undoes the flip, exactly
        if (rec) { const f4* tg = sv.shade_tangents + 3u * record; g0 = tg[0]; g1 = tg[1]; g2 = tg[2]; }
        const f4 avg_tangent = g0 * b.x + g1 * b.y + g2 * b.z;
        const f3 front = back_facing ? -smooth_normal : smooth_normal;
        const f3 tangent = normalize(pre ? F3(avg_tangent) : mul(mn, F3(avg_tangent)));
        const f3 bitangent = normalize(cross(front, tangent) * avg_tangent.w);
        m3 tbn = {{tangent, bitangent, smooth_normal}};
        f4 t = sample_texture(sv, mat.normal_tex_id, uv);
        f3 ts_normal = normalize(F3(t) * 2.0f - 1.0f);
        f3 mapped = normalize(mul(tbn, ts_normal * F3(mat.normal_factor, mat.normal_factor, 1.0f)));
        sp.mapped_normal = any_nan(mapped) ? smooth_normal : mapped;
    }
    STL(STL_NORMALMAP);
    res.emission = F3(mat.emission_factor);
    if (mat.emission_tex_id >= 0) res.emission = res.emission * F3(sample_texture(sv, mat.emission_tex_id, uv));
    res.transmittance = mat.transmittance;
    if (back_facing && res.transmittance > 0.0001f) { res.ior_in = mat.ior; res.ior_out = 1.0f; }
    else { res.ior_in = 1.0f; res.ior_out = mat.ior; }
    float f0 = (res.ior_out - res.ior_in) / (res.ior_out + res.ior_in);
    res.f0 = f0 * f0;
    STL(STL_EMISSION);
}

}  // namespace tr
