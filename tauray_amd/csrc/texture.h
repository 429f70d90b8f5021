// Texture sampling for ray-tracing stages: implicit-LOD texture() has no derivatives there, so it is
// base level, bilinear, repeat addressing (src/sampler_table.cc:8-17), on un-normalised RGBA8 texels or - a 16-bit PNG, which the
// reference keeps as R16G16B16A16Unorm (src/gltf.cc:548-556) - RGBA16 texels (TextureInfo::format).
#pragma once
#include "common.h"

namespace tr {

TR_DEV int wrap_repeat(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }

TR_DEV f4 fetch_rgba8(const uint8_t* base, int w, int x, int y) {
    uint p = *reinterpret_cast<const uint*>(base + ((size_t)y * (size_t)w + (size_t)x) * 4);   // one dword per texel
    return F4((float)(p & 0xFF), (float)((p >> 8) & 0xFF), (float)((p >> 16) & 0xFF), (float)(p >> 24)) * (1.0f / 255.0f);
}

// One texel of either format as four floats in [0, 1]; `wide` = TEXTURE_FORMAT_RGBA16 (two dwords per texel instead of one).
TR_DEV f4 fetch_texel(const uint8_t* base, int w, int x, int y, uint wide) {
    const size_t idx = (size_t)y * (size_t)w + (size_t)x;
    const uint d0 = *reinterpret_cast<const uint*>(base + (idx << (2u + wide)));
    const uint d1 = wide ? *reinterpret_cast<const uint*>(base + (idx << 3) + 4) : d0;
    const uint m = wide ? 0xFFFFu : 0xFFu;
    const float scale = wide ? (1.0f / 65535.0f) : (1.0f / 255.0f);
    return F4((float)(d0 & m), (float)((d0 >> (wide ? 16u : 8u)) & m), (float)((wide ? d1 : (d0 >> 16)) & m), (float)(wide ? (d1 >> 16) : (d0 >> 24))) * scale;
}
// ... and its alpha alone, for the any-hit test of the traversal (candidate_alpha, trace.h): the same value as fetch_texel(...).w
TR_DEV float fetch_alpha(const uint8_t* base, int w, int x, int y, uint wide) {
    const size_t idx = (size_t)y * (size_t)w + (size_t)x;
    const uint d = *reinterpret_cast<const uint*>(base + (idx << (2u + wide)) + (wide << 2));
    return (float)(d >> (wide ? 16u : 24u)) * (wide ? (1.0f / 65535.0f) : (1.0f / 255.0f));
}

TR_DEV f4 bilerp(f4 c00, f4 c10, f4 c01, f4 c11, float fx, float fy) {
    f4 top = c00 * (1.0f - fx) + c10 * fx;
    f4 bot = c01 * (1.0f - fx) + c11 * fx;
    return top * (1.0f - fy) + bot * fy;
}

// texel coordinates and weights of a bilinear tap with repeat addressing
struct BilinearTap { int x0, y0, x1, y1; float fx, fy; };
TR_DEV BilinearTap bilinear_tap(int w, int h, f2 uv) {
    // a texture unit returns a texel for any coordinate; non-finite ones (the uv of a light sample on a triangle seen
    // edge-on, whose weight is zero anyway) are defined as 0 here so that 0 * texel stays 0
    if (!(fabsf(uv.x) < __builtin_huge_valf())) uv.x = 0.0f;
    if (!(fabsf(uv.y) < __builtin_huge_valf())) uv.y = 0.0f;
    float x = uv.x * (float)w - 0.5f, y = uv.y * (float)h - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    BilinearTap t;
    t.fx = x - fx0; t.fy = y - fy0;
    t.x0 = wrap_repeat((int)fx0, w); t.y0 = wrap_repeat((int)fy0, h);
    t.x1 = wrap_repeat((int)fx0 + 1, w); t.y1 = wrap_repeat((int)fy0 + 1, h);
    return t;
}

// A scene without RGBA16 textures - nearly every scene - takes the one-dword code the kernels had before the format existed: the
// branch is on a kernel argument (scalar), so that path costs nothing and keeps its registers (an RGBA16 branch per texel, or selects
// on every fetch, cost the trace kernels 4-5 % and the pipelined frame 8 %: profiles/r4/texture_format_ab.txt).
TR_DEV f4 sample_texture_rgba8(const SceneView& sv, const TextureInfo ti, f2 uv) {
    int w = (int)ti.width, h = (int)ti.height;
    if (!(fabsf(uv.x) < __builtin_huge_valf())) uv.x = 0.0f;
    if (!(fabsf(uv.y) < __builtin_huge_valf())) uv.y = 0.0f;
    float x = uv.x * (float)w - 0.5f, y = uv.y * (float)h - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int x0 = wrap_repeat((int)fx0, w), y0 = wrap_repeat((int)fy0, h);
    int x1 = wrap_repeat((int)fx0 + 1, w), y1 = wrap_repeat((int)fy0 + 1, h);
    const uint8_t* base = sv.texels + (size_t)ti.texel_offset * 4;
    return bilerp(fetch_rgba8(base, w, x0, y0), fetch_rgba8(base, w, x1, y0), fetch_rgba8(base, w, x0, y1),
                  fetch_rgba8(base, w, x1, y1), fx, fy);
}
TR_DEV f4 sample_texture_any(const SceneView& sv, const TextureInfo ti, f2 uv) {
    const int w = (int)ti.width, h = (int)ti.height;
    const BilinearTap t = bilinear_tap(w, h, uv);
    const uint8_t* base = sv.texels + (size_t)ti.texel_offset * 4;
    const uint wide = ti.format == TEXTURE_FORMAT_RGBA16 ? 1u : 0u;
    return bilerp(fetch_texel(base, w, t.x0, t.y0, wide), fetch_texel(base, w, t.x1, t.y0, wide), fetch_texel(base, w, t.x0, t.y1, wide),
                  fetch_texel(base, w, t.x1, t.y1, wide), t.fx, t.fy);
}
TR_DEV f4 sample_texture(const SceneView& sv, int tex_id, f2 uv) {
    const TextureInfo ti = sv.tex_infos[tex_id];
    if (!sv.wide_textures) return sample_texture_rgba8(sv, ti, uv);
    return sample_texture_any(sv, ti, uv);
}
// sample_texture(...).w for the any-hit test of the traversal (candidate_alpha, trace.h)
TR_DEV float sample_texture_alpha(const SceneView& sv, int tex_id, f2 uv) {
    const TextureInfo ti = sv.tex_infos[tex_id];
    if (!sv.wide_textures) return sample_texture_rgba8(sv, ti, uv).w;
    const int w = (int)ti.width, h = (int)ti.height;
    const BilinearTap t = bilinear_tap(w, h, uv);
    const uint8_t* base = sv.texels + (size_t)ti.texel_offset * 4;
    const uint wide = ti.format == TEXTURE_FORMAT_RGBA16 ? 1u : 0u;
    const float a00 = fetch_alpha(base, w, t.x0, t.y0, wide), a10 = fetch_alpha(base, w, t.x1, t.y0, wide), a01 = fetch_alpha(base, w, t.x0, t.y1, wide),
                a11 = fetch_alpha(base, w, t.x1, t.y1, wide);
    const float top = a00 * (1.0f - t.fx) + a10 * t.fx, bot = a01 * (1.0f - t.fx) + a11 * t.fx;
    return top * (1.0f - t.fy) + bot * t.fy;
}

// i mod n for a coordinate one period away at most: what a uv in [0, 1] produces (texel -1 ... n), i.e. every environment lookup - the uv of
// a direction (asin / atan2 over pi) or of an alias-table sample.  A 32-bit remainder is ~35 instructions here (no integer divider), an
// environment tap used four, and a shade wave spends a fifth of its clocks in the two environment lookups of a bounce
// (profiles/r6/shade_phase_timeline.txt).  Any other coordinate takes the remainder as before: the same integer either way.
TR_DEV int wrap_repeat_near(int i, int n) {
    if (__builtin_expect((uint)(i + n) < 3u * (uint)n, 1)) return i < 0 ? i + n : (i >= n ? i - n : i);
    return wrap_repeat(i, n);
}
TR_DEV f4 sample_envmap(const SceneView& sv, f2 uv) {
    int w = (int)sv.env_w, h = (int)sv.env_h;
    float x = uv.x * (float)w - 0.5f, y = uv.y * (float)h - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int x0 = wrap_repeat_near((int)fx0, w), y0 = wrap_repeat_near((int)fy0, h);
    int x1 = wrap_repeat_near((int)fx0 + 1, w), y1 = wrap_repeat_near((int)fy0 + 1, h);
    const f4* e = sv.envmap;
    return bilerp(e[(size_t)y0 * w + x0], e[(size_t)y0 * w + x1], e[(size_t)y1 * w + x0], e[(size_t)y1 * w + x1], fx, fy);
}

}  // namespace tr
